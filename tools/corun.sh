#!/bin/bash
# run a command while a SECOND process keeps the same GPU busy (matrix products on its own context): the perturbation that exposes
# ordering bugs between the two lanes of the training step.  usage: tools/corun.sh <command ...>
python - <<'PY' &
import torch, time
d = torch.device("cuda:0")
a = torch.randn(2048, 2048, device=d)
t0 = time.time()
while time.time() - t0 < 600:
    for _ in range(20):
        a = (a @ a).clamp_(-1, 1)
    torch.cuda.synchronize()
    time.sleep(0.002)
PY
bg=$!
"$@"
rc=$?
kill $bg 2>/dev/null
wait $bg 2>/dev/null
exit $rc
