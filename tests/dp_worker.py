"""Worker of tests/test_dp_product_gpu.py: ONE rank of a world-2 data-parallel step of the PRODUCT generator on cuda:0.
Both ranks share the one GPU of the test box; the collectives run on the backend named in DP_BACKEND (gloo on device tensors, or
nccl = RCCL if it accepts two ranks on one device).  Rank r builds TSRN_TL_TRANS from seed 1234 + 1000 r (so rank 1's initial weights
DIFFER from rank 0's and the start-up broadcast is observable), trains one step on its own shard of the batch and dumps what the
parent test checks."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend, out_dir = os.environ["DP_BACKEND"], os.environ["DP_OUT"]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    kw = dict(backend=backend, init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank, world_size=world)
    if backend == "nccl":
        kw["device_id"] = dev
    dist.init_process_group(**kw)
    import tatt_amd
    from oracle.fixtures import randomize_state_dict, make_inputs
    from tatt_amd import ops
    from tatt_amd.train import Trainer
    STD = dict(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)
    torch.manual_seed(1234 + 1000 * rank)
    m = tatt_amd.TSRN_TL_TRANS(**STD)
    m.load_state_dict(randomize_state_dict(m.state_dict(), seed=rank))          # rank-dependent: only the broadcast makes them equal
    m = m.to(dev).train()
    m.infoGen.dropout_on = False
    w_before = m.block2.conv1.weight.detach().clone()
    tr = Trainer(m, use_graph=False, process_group=dist.group.WORLD)       # (broadcasts rank 0's weights and buffers)
    w = m.block2.conv1.weight
    w_changed = bool((w_before != w).any().item())
    # Packed filter copies are cached per parameter and keyed on torch's version counter; a broadcast writes through the flat buffer
    # behind that counter, so broadcast_model must rebuild them.  Cache one, move rank 0's weights, broadcast again:
    from tatt_amd.dp import broadcast_model
    packed0 = ops.repack_weight(w, 10)
    p0 = packed0.clone()
    if rank == 0:
        tr.flat.p.mul_(1.001)
    broadcast_model(tr.flat, m, dist.group.WORLD)
    packed1 = ops.repack_weight(w, 10)                                      # answered by the cache (same buffer) ...
    fresh = torch.empty_like(packed1)
    ops.call("tatt_repack_conv_weight", ops.P(w), ops.P(fresh), 64, 64, 3, 3, 10, ops.stream())
    packed_same_buffer = packed1.data_ptr() == packed0.data_ptr()
    packed_changed = bool((p0 != packed1).any().item())
    packed_fresh = bool(torch.equal(packed1, fresh))                        # ... which holds the NEW weights' layout
    sd_start = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x, tp, hr = make_inputs(2 * world, seed=60)
    sl = slice(2 * rank, 2 * rank + 2)
    loss = tr.step(x[sl].to(dev), tp[sl].to(dev), hr[sl].to(dev))
    torch.cuda.synchronize()
    n = tr.flat.n
    res = {
        "loss": float(loss), "grad_norm": float(tr.last_grad_norm), "reduce_log": list(tr.reduce_log), "stages": list(tr.stages),
        "world": dist.get_world_size(), "backend": dist.get_backend(),
        "w_changed_by_broadcast": w_changed, "packed_same_buffer": packed_same_buffer,
        "packed_changed": packed_changed, "packed_fresh": packed_fresh,
        "sd_start": sd_start, "sd_end": {k: v.detach().cpu().clone() for k, v in m.state_dict().items()},
        "m": {k: tr.flat_m[o:o + c].detach().cpu().clone().view_as(p) for (k, p) in m.named_parameters() for (o, c) in [tr.flat.offsets[id(p)]]},
        "v": {k: tr.flat_v[o:o + c].detach().cpu().clone().view_as(p) for (k, p) in m.named_parameters() for (o, c) in [tr.flat.offsets[id(p)]]},
        "n_flat": n,
    }
    torch.save(res, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    main()
