"""Output side of the harness: decoded clips -> files (second half).

Counterpart of the reference's `save_results` / `save_results_seperate` (scripts/evaluation/inference.py:115-162) and
`tensor_to_mp4` (utils/save_video.py:27-43). Their arithmetic - clamp to [-1,1], (v+1)/2, x255, uint8 truncation, the
n clips of a batch side by side (`make_grid(nrow=n, padding=0)`), frames as [t, h, w, c] - is one HIP kernel
(`dc_frames_to_u8`) on the decoded tensor where it lies. The container differs: the reference hands the frames to
torchvision.io.write_video (h264, crf 10); no video encoder exists in this image, so clips are written as APNG
(animated PNG: lossless, zlib only, one file per clip, plays in browsers) and single frames as PNG. File names keep the
reference's stems; only the extension changes (.png instead of .mp4).
"""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import torch

from .. import _hip
from ..ops import stream_ptr


def frames_to_uint8(video):
    """video [n, c, t, h, w] fp32 on the GPU, values nominally in [-1, 1] -> uint8 [t, h, n*w, c] (same device)."""
    if not video.is_cuda:
        raise RuntimeError("frames_to_uint8 runs on the HIP path only (there is no CPU fallback)")
    if video.dim() != 5:
        raise ValueError(f"expected [n, c, t, h, w], got {tuple(video.shape)}")
    v = video.detach().to(torch.float32).contiguous()
    n, c, t, h, w = v.shape
    out = torch.empty((t, h, n * w, c), dtype=torch.uint8, device=v.device)
    _hip.check(_hip.lib().dc_frames_to_u8(C.c_void_p(v.data_ptr()), C.c_void_p(out.data_ptr()), n, c, t, h, w, stream_ptr()),
               "dc_frames_to_u8")
    return out


# ---------------------------------------------------------------------------------------------- PNG / APNG (zlib only)
def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _ihdr(w, h, c):
    color = {1: 0, 3: 2, 4: 6}[c]
    return _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0))


def _scanlines(frame, level):
    """uint8 [h, w, c] -> zlib stream of filter-0 scanlines"""
    h = frame.shape[0]
    raw = np.empty((h, 1 + frame.shape[1] * frame.shape[2]), dtype=np.uint8)
    raw[:, 0] = 0
    raw[:, 1:] = frame.reshape(h, -1)
    return zlib.compress(raw.tobytes(), level)


def write_png(path, frame, level=6):
    """frame: uint8 [h, w, c] (c in 1, 3, 4), numpy or tensor."""
    f = np.ascontiguousarray(frame.cpu().numpy() if isinstance(frame, torch.Tensor) else frame)
    if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] not in (1, 3, 4):
        raise ValueError(f"write_png: uint8 [h, w, 1|3|4] expected, got {f.dtype} {f.shape}")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + _ihdr(f.shape[1], f.shape[0], f.shape[2]) + _chunk(b"IDAT", _scanlines(f, level))
                 + _chunk(b"IEND", b""))
    return path


def write_apng(path, frames, fps=8, level=6, loops=0):
    """frames: uint8 [t, h, w, c]. Animated PNG (acTL / fcTL / fdAT): frame 0 doubles as the still image every PNG
    reader shows; `loops` = 0 repeats forever."""
    f = np.ascontiguousarray(frames.cpu().numpy() if isinstance(frames, torch.Tensor) else frames)
    if f.dtype != np.uint8 or f.ndim != 4 or f.shape[3] not in (1, 3, 4):
        raise ValueError(f"write_apng: uint8 [t, h, w, 1|3|4] expected, got {f.dtype} {f.shape}")
    t, h, w, c = f.shape
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    seq = 0
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + _ihdr(w, h, c) + _chunk(b"acTL", struct.pack(">II", t, loops)))
        for i in range(t):
            fh.write(_chunk(b"fcTL", struct.pack(">IIIIIHHBB", seq, w, h, 0, 0, 1, int(fps), 0, 0)))
            seq += 1
            data = _scanlines(f[i], level)
            if i == 0:
                fh.write(_chunk(b"IDAT", data))
            else:
                fh.write(_chunk(b"fdAT", struct.pack(">I", seq) + data))
                seq += 1
        fh.write(_chunk(b"IEND", b""))
    return path


# ---------------------------------------------------------------------------------------------- reference-named entry points
def save_results(prompt, samples, filename, fakedir, fps=8, loop=False):
    """inference.py:115-137: the batch as ONE clip, its n samples side by side. samples [n, c, t, h, w]."""
    video = samples[:, :, :-1] if loop else samples            # loop mode drops the duplicated last frame
    grid = frames_to_uint8(video)
    return write_apng(os.path.join(fakedir, filename.split(".")[0] + ".png"), grid, fps=fps)


def save_results_seperate(prompt, samples, filename, fakedir, fps=10, loop=False):
    """inference.py:140-162: one clip file per sample, under `samples_separate` (name kept as the reference spells it)."""
    video = samples[:, :, :-1] if loop else samples
    out = []
    d = fakedir.replace("samples", "samples_separate")
    for i in range(video.shape[0]):
        grid = frames_to_uint8(video[i:i + 1])
        out.append(write_apng(os.path.join(d, f"{filename.split('.')[0]}_sample{i}.png"), grid, fps=fps))
    return out


def tensor_to_frames(video, savedir, stem="frame"):
    """One PNG per frame of a [n, c, t, h, w] batch laid out side by side (the still-image twin of tensor_to_mp4,
    utils/save_video.py:27-43)."""
    grid = frames_to_uint8(video).cpu().numpy()
    return [write_png(os.path.join(savedir, f"{stem}_{i:04d}.png"), grid[i]) for i in range(grid.shape[0])]
