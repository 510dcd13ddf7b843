"""Checkpoint files and the evaluation metrics loop in the reference's formats.

* `save_checkpoint` writes what `TextBase.save_checkpoint` writes (reference interfaces/base.py:621-672): one file per generator
  with the dict {'state_dict_G', 'info', 'best_history_res', 'best_model_info', 'param_num', 'converge'} as
  `model_best_<prefix>_<i>.pth` / `checkpoint.pth`, the recognisers' state_dicts beside them.
* `load_generator` mirrors the resume branch of `TextBase.generator_init` (reference interfaces/base.py:398-443): a file or a
  directory, with or without the 'state_dict_G' wrapper, `module.`-prefixed keys of DataParallel checkpoints accepted in both
  directions.
* `evaluate` is the metric part of the reference's eval loop (interfaces/super_resolution.py:1409-1420,1454-1455): PSNR and SSIM
  of the SR images against HR on the first three channels, averaged over batches -- computed by the HIP kernels -- and, given a
  CRNN recogniser and the label strings, the recognition accuracies of the SR / LR / HR images (:1374-1396,1527-1558,1662-1664;
  greedy CTC decoding = utils/metrics.py:71-92, string filter = utils/util.py:12-32).
* `collate_labels` / `collate_batch` produce the batch tuple the reference's loaders hand to the loop
  (dataset/dataset.py:1966-2077, alignCollate_realWTLAMask.__call__): images stacked, labels stretched to 26 steps and one-hot
  encoded as the (B, 37, 1, 26) text prior, the per-character class list and the blank flags.
* `resize_normalize` / `collate_pil_batch` are the image half of that collate (`resizeNormalize`, dataset/dataset.py:1266-1319; the
  transform calls of :1987-2003): PIL bicubic resize to the HR / LR size, uint8 -> float / 255 in CHW, and the binarised mask channel
  (gray < mean -> 1) as the fourth plane -- the (B, 4, H, W) tensors the generator reads.  Pinned by `tests/golden/collate.npz`,
  generated from the reference's own collate (tools/gen_golden_collate.py).
* `LmdbRecords` reads the reference's lmdb record layout (`lmdbDataset_real`, dataset/dataset.py:565-686): keys `num-samples`,
  `label-%09d`, `image_hr-%09d`, `image_lr-%09d` (1-based), image bytes decoded by PIL to RGB, the label filtered by `str_filt`.
  It takes any object with the lmdb transaction's `get(key)`; `open_lmdb` wraps a real environment when the `lmdb` package is there
  (it is not in this image: the record logic is tested against an in-memory mapping).
Host-side data plumbing (PIL / numpy), not part of the GPU path: torch.save / torch.load / PIL are file-format code; all arithmetic of
the training step stays in the HIP kernels.
"""
from __future__ import annotations

import os
import string
from typing import Iterable, Optional, Sequence

import torch

ALPHABET = "0123456789abcdefghijklmnopqrstuvwxyz"          # class 0 is the CTC blank "-" (reference utils/metrics.py:71)


def _unwrap(m):
    return m.module if hasattr(m, "module") and isinstance(getattr(m, "module"), torch.nn.Module) else m


def save_checkpoint(netG_list: Sequence[torch.nn.Module], epoch: int, iters: int, best_acc_dict, best_model_info, is_best: bool,
                    converge_list, ckpt_path: str, *, arch: str = "tatt", batch_size: int = 48, voc_type: str = "all",
                    scale_factor: int = 2, recognizer=None, prefix: str = "acc"):
    """reference TextBase.save_checkpoint (interfaces/base.py:621-672); returns the list of files written."""
    os.makedirs(ckpt_path, exist_ok=True)
    written = []
    for i, net in enumerate(netG_list):
        netG = _unwrap(net)
        save_dict = {
            "state_dict_G": {k: v.detach().cpu() for k, v in netG.state_dict().items()},
            "info": {"arch": arch, "iters": iters, "epochs": epoch, "batch_size": batch_size, "voc_type": voc_type,
                     "up_scale_factor": scale_factor},
            "best_history_res": best_acc_dict,
            "best_model_info": best_model_info,
            "param_num": sum(p.nelement() for p in netG.parameters()),
            "converge": converge_list,
        }
        name = ("model_best_%s_%d.pth" % (prefix, i)) if is_best else "checkpoint.pth"
        torch.save(save_dict, os.path.join(ckpt_path, name))
        written.append(os.path.join(ckpt_path, name))
    if recognizer is not None:
        recs = recognizer if isinstance(recognizer, (list, tuple)) else [recognizer]
        for i, r in enumerate(recs):
            if isinstance(recognizer, (list, tuple)):
                name = ("recognizer_best_%s_%d.pth" % (prefix, i)) if is_best else "recognizer_%d.pth" % i
            else:
                name = "recognizer_best.pth" if is_best else "recognizer.pth"
            torch.save({k: v.detach().cpu() for k, v in _unwrap(r).state_dict().items()}, os.path.join(ckpt_path, name))
            written.append(os.path.join(ckpt_path, name))
    return written


def _strip_module(sd):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_generator(model: torch.nn.Module, resume: str, iter_: int = 0, strict: Optional[bool] = None):
    """reference TextBase.generator_init resume branch (interfaces/base.py:398-443).  `resume` is a checkpoint file or a directory
    holding model_best_acc_<iter_>.pth; directories load with strict=False like the reference, files strictly.  Returns the
    checkpoint's 'info' dict (or None for a bare state_dict)."""
    is_dir = os.path.isdir(resume)
    path = os.path.join(resume, "model_best_acc_%d.pth" % iter_) if is_dir else resume
    blob = torch.load(path, map_location="cpu")
    sd = blob["state_dict_G"] if isinstance(blob, dict) and "state_dict_G" in blob else blob
    target = _unwrap(model)
    target.load_state_dict(_strip_module(sd), strict=(not is_dir) if strict is None else strict)
    return blob.get("info") if isinstance(blob, dict) and "state_dict_G" in blob else None


def stretch_label(word: str, max_len: int = 26) -> str:
    """The reference's label layout (dataset/dataset.py:2015-2034): lower-cased; words of 2..25 characters are spread over the
    26 prior steps by inserting int((26 - len) / (len - 1)) blanks between neighbouring characters; longer words are cut."""
    word = word.lower()
    if len(word) <= 1:
        return word
    if len(word) < max_len:
        pad = int((max_len - len(word)) / (len(word) - 1))
        return word[0] + "".join("-" * pad + ch for ch in word[1:])
    return word[:max_len]


def collate_labels(label_strs: Sequence[str], alphabet: str = ALPHABET, max_len: int = 26):
    """-> (label_vecs (B, len(alphabet)+1, 1, max_len) one-hot float, weighted_mask (sum of label lengths,) long, weighted_tics (B,) long)
    exactly as alignCollate_realWTLAMask builds them (dataset/dataset.py:2009-2063): characters outside the alphabet are dropped, a
    word with no valid character becomes a single blank (class 0) with tic 0."""
    d2a = "-" + alphabet
    a2d = {ch: i for i, ch in enumerate(d2a)}
    alsize = len(d2a)
    out = torch.zeros(len(label_strs), max_len, alsize)
    masks, tics = [], []
    for b, word in enumerate(label_strs):
        ids = [a2d[ch] for ch in stretch_label(word, max_len) if ch in a2d]
        if ids:
            masks.extend(ids)
            out[b, torch.arange(len(ids)), torch.tensor(ids)] = 1.0
            tics.append(1)
        else:
            masks.append(0)
            out[b, 0, 0] = 1.0
            tics.append(0)
    return out.unsqueeze(1).permute(0, 3, 1, 2).contiguous(), torch.tensor(masks).long(), torch.tensor(tics)


def collate_batch(samples, device=None, alphabet: str = ALPHABET):
    """samples: iterable of (img_HR, img_lr, img_HRy, img_lry, label_str) with the images already tensors of their final size
    ((4 or 3, H, W) float in [0, 1]) -> the tuple of the reference's collate function (dataset/dataset.py:2077):
    (images_HR, images_pseudoLR = None, images_lr, images_HRy, images_lry, label_strs, label_vecs, weighted_mask, weighted_tics),
    image stacks and label_vecs on `device` (the loop's `.to(self.device)`, interfaces/super_resolution.py:700-707)."""
    hr, lr, hry, lry, labels = zip(*samples)
    st = lambda ts: torch.stack([torch.as_tensor(t) for t in ts], 0).to(device) if device is not None else torch.stack(
        [torch.as_tensor(t) for t in ts], 0)
    vecs, masks, tics = collate_labels(labels, alphabet)
    return st(hr), None, st(lr), st(hry), st(lry), tuple(labels), (vecs.to(device) if device is not None else vecs), masks, tics


def _to_tensor(img):
    """torchvision.transforms.ToTensor for a uint8 PIL image: (H, W[, C]) uint8 -> (C, H, W) float32 in [0, 1]"""
    import numpy as np
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.array(a.transpose(2, 0, 1))).float().div(255)                  # (np.array: a writable, contiguous copy)


def resize_normalize(img, size, mask: bool = False):
    """reference `resizeNormalize(size, mask)(img)` (dataset/dataset.py:1266-1319, the ratio_keep / aug branches unused by the TATT
    loaders): img: PIL image; size = (width, height).  Bicubic resize, ToTensor, and with `mask` a fourth plane that is 1 where the
    gray value does not exceed the image's mean gray value (`mask.point(lambda x: 0 if x > thres else 255)`) -- dark text on a
    bright background comes out as 1."""
    import numpy as np
    from PIL import Image
    img = img.resize(tuple(size), Image.BICUBIC)
    t = _to_tensor(img)
    if mask:
        m = img.convert("L")
        thres = np.array(m).mean()
        m = m.point(lambda v: 0 if v > thres else 255)
        t = torch.cat((t, _to_tensor(m)), 0)
    return t


def collate_pil_batch(samples, imgH: int = 32, imgW: int = 128, down_sample_scale: int = 2, mask: bool = True, device=None,
                      alphabet: str = ALPHABET):
    """samples: iterable of (img_HR, img_lr, img_HRy, img_lry, label_str) with PIL images, as `lmdbDataset_real.__getitem__` yields
    them -> the reference's batch tuple (alignCollate_realWTLAMask.__call__, dataset/dataset.py:1980-2077): HR images resized to
    (imgW, imgH), LR images to (imgW, imgH) / down_sample_scale, each with its mask plane."""
    hr_size, lr_size = (imgW, imgH), (imgW // down_sample_scale, imgH // down_sample_scale)
    rows = [(resize_normalize(hr, hr_size, mask), resize_normalize(lr, lr_size, mask), resize_normalize(hry, hr_size, mask),
             resize_normalize(lry, lr_size, mask), lab) for hr, lr, hry, lry, lab in samples]
    return collate_batch(rows, device=device, alphabet=alphabet)


def rgb_to_yuv_u8(rgb):
    """cv2.cvtColor(img, cv2.COLOR_RGB2YUV) for uint8 (dataset/dataset.py:668-674: the `images_lry` / `images_HRy` members of a sample,
    read by the loop only with --y_domain, which the TATT recipes do not set): Y = 0.299 R + 0.587 G + 0.114 B, U = 0.492 (B - Y) + 128,
    V = 0.877 (R - Y) + 128, rounded and saturated.  (cv2 evaluates this in 14-bit fixed point; results may differ from it by one
    count -- cv2 is not in this image, so this member is NOT pinned against the reference.)"""
    import numpy as np
    a = np.asarray(rgb).astype(np.float64)
    y = 0.299 * a[..., 0] + 0.587 * a[..., 1] + 0.114 * a[..., 2]
    u = 0.492 * (a[..., 2] - y) + 128.0
    v = 0.877 * (a[..., 0] - y) + 128.0
    return np.clip(np.rint(np.stack([y, u, v], -1)), 0, 255).astype(np.uint8)


class LmdbRecords:
    """The reference's lmdb record layout read through a transaction-like object (`get(bytes) -> bytes or None`), reference
    `lmdbDataset_real` (dataset/dataset.py:565-686): `num-samples` holds the count, sample i (1-based) is `image_hr-%09d`,
    `image_lr-%09d` (encoded image files) and `label-%09d` (utf-8).  __getitem__(index) -> (img_HR, img_lr, img_HRy, img_lry,
    label_str) with PIL images, the label passed through `str_filt(word, voc_type)`.
    Bad records, exactly as the reference behaves (dataset/dataset.py:640-686): its `except IOError or len(word) > self.max_len` catches
    IOError ONLY (the `or` picks the class), so an over-long label is RETURNED, not skipped (`max_len` is kept as an attribute, unused as
    upstream); on an unreadable image it returns `self[index + 1]` AFTER `index += 1`, i.e. 0-based item i falls through to item i + 2 --
    one record further than it looks.  Running past the last record raises IndexError (upstream: an assertion / a missing key).
    NOT exercised against a real lmdb environment: the `lmdb` package is absent from this image (tests use an in-memory mapping)."""

    def __init__(self, txn, voc_type: str = "upper", max_len: int = 100):
        self.txn, self.voc_type, self.max_len = txn, voc_type, max_len
        n = txn.get(b"num-samples")
        if n is None:
            raise KeyError("lmdb environment without a num-samples record")
        self.n = int(n)

    def __len__(self):
        return self.n

    def _image(self, key):
        import io as _io
        from PIL import Image
        buf = self.txn.get(key)
        if buf is None:
            raise IOError("missing record %r" % key)
        return Image.open(_io.BytesIO(buf)).convert("RGB")

    def __getitem__(self, index):
        from PIL import Image
        if not 0 <= index < self.n:
            raise IndexError(index)
        while index < self.n:
            i = index + 1                                            # 1-based record
            try:
                hr, lr = self._image(b"image_hr-%09d" % i), self._image(b"image_lr-%09d" % i)
            except (IOError, OSError):
                index += 2                                           # the reference's `return self[index + 1]` after `index += 1`
                continue
            word = self.txn.get(b"label-%09d" % i)
            word = " " if word is None else word.decode()
            hry, lry = Image.fromarray(rgb_to_yuv_u8(hr)), Image.fromarray(rgb_to_yuv_u8(lr))
            return hr, lr, hry, lry, str_filt(word, self.voc_type)
        raise IndexError("no readable record at or after the requested index")


def open_lmdb(root: str, **kw) -> LmdbRecords:
    """`lmdbDataset_real(root)`: opens the environment read-only like the reference (dataset/dataset.py:576-583).  Needs the `lmdb`
    package (absent from this image: ImportError says so)."""
    import lmdb                                                  # noqa: F401 -- optional dependency of the data pipeline only
    env = lmdb.open(root, max_readers=1, readonly=True, lock=False, readahead=False, meminit=False)
    return LmdbRecords(env.begin(write=False), **kw)


def ctc_greedy_decode(logits: torch.Tensor, alphabet: str = ALPHABET) -> list:
    """(T, B, C) recogniser outputs -> B strings: arg-max per step, repeats merged, blanks dropped (reference get_string_crnn,
    utils/metrics.py:71-92).  The arg-max is an index operation on a (T, B) grid; the strings are built on the host."""
    d2a = "-" + alphabet
    idx = logits.detach().permute(1, 0, 2).argmax(2).cpu().tolist()
    out = []
    for row in idx:
        s, last = "", ""
        for i in row:
            if d2a[i] != last:
                if i != 0:
                    s += d2a[i]
                    last = d2a[i]
                else:
                    last = ""
        out.append(s)
    return out


def str_filt(s: str, voc_type: str = "lower") -> str:
    """reference utils/util.py:12-32 for the Latin vocabularies ('digit', 'lower', 'upper', 'all')."""
    alpha = {"digit": string.digits, "lower": string.digits + string.ascii_lowercase, "upper": string.digits + string.ascii_letters,
             "all": string.digits + string.ascii_letters + string.punctuation}[voc_type]
    if voc_type == "lower":
        s = s.lower()
    return "".join(ch for ch in s if ch in alpha)


@torch.no_grad()
def evaluate(model: torch.nn.Module, batches: Iterable, prior_fn=None, recognizer=None, voc_type: str = "lower"):
    """Metric part of the reference's eval loop: for every (images_lr, images_hr[, text_prior[, label_strs]]) batch run the
    generator in eval mode and accumulate calculate_psnr / SSIM of SR vs HR on the first three channels
    (interfaces/super_resolution.py:1454-1455).  With a `recognizer` (tatt_amd.CRNN, the reference's --test_model CRNN) and label
    strings in the batch, also the recognition accuracies of the SR, LR and HR images (:1374-1396,1527-1558,1662-1664).
    Returns {'psnr', 'ssim', 'n_batches'[, 'accuracy', 'accuracy_lr', 'accuracy_hr', 'n_images']} (python floats)."""
    from .losses import SSIM, calculate_psnr
    from .crnn import parse_crnn_data
    was_training = model.training
    model.eval()
    ssim = SSIM()
    psnr_sum = torch.zeros((), device=next(model.parameters()).device)
    ssim_sum = torch.zeros_like(psnr_sum)
    n = 0
    correct = {"sr": 0, "lr": 0, "hr": 0}
    n_img = 0
    rec_was_training = recognizer.training if recognizer is not None else False
    if recognizer is not None:
        recognizer.eval()
    for batch in batches:
        lr, hr = batch[0], batch[1]
        tp = batch[2] if len(batch) > 2 and batch[2] is not None else (prior_fn(lr) if prior_fn is not None else None)
        labels = batch[3] if len(batch) > 3 else None
        out = model(lr, tp) if tp is not None else model(lr)
        sr = out[0] if isinstance(out, tuple) else out
        psnr_sum += calculate_psnr(sr[:, :3], hr[:, :3])
        ssim_sum += ssim(sr[:, :3], hr[:, :3])
        n += 1
        if recognizer is not None and labels is not None:
            for name, img in (("sr", sr), ("lr", lr), ("hr", hr)):
                pred = ctc_greedy_decode(recognizer(parse_crnn_data(img[:, :3].contiguous())))
                correct[name] += sum(str_filt(p, voc_type) == str_filt(t, voc_type) for p, t in zip(pred, labels))
            n_img += len(labels)
    model.train(was_training)
    if recognizer is not None:
        recognizer.train(rec_was_training)               # e.g. TextPriorSR.tpg under training: BatchNorm must not stay in eval mode
    res = {"psnr": float(psnr_sum) / max(n, 1), "ssim": float(ssim_sum) / max(n, 1), "n_batches": n}
    if n_img:
        res.update(accuracy=round(correct["sr"] / n_img, 4), accuracy_lr=round(correct["lr"] / n_img, 4),
                   accuracy_hr=round(correct["hr"] / n_img, 4), n_images=n_img)
    return res
