"""Audio file loading without torchaudio (absent from the image): FLAC through the native decoder of
libcpc2_hip.so (cpc_flac_decode_f32, MD5-verified), PCM / float WAV through a small RIFF parser.
`load(path)` returns what `torchaudio.load(path)[0]` returns: float32 [channels, samples] in [-1, 1)
(reference call sites: cpc/dataset.py:411-437, cpc/feature_loader.py:343)."""
import ctypes
import struct

import numpy as np
import torch

from . import _lib


def info(path):
    """(sample_rate, channels, num_frames) -- torchaudio.info(...).num_frames of dataset.py:759-768."""
    path = str(path)
    if path.lower().endswith(".flac"):
        sr, ch, bps, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_long()
        _lib.check(_lib.load().cpc_flac_info(path.encode(), ctypes.byref(sr), ctypes.byref(ch), ctypes.byref(bps),
                                             ctypes.byref(n)), f"flac_info({path})")
        return sr.value, ch.value, n.value
    wav, sr = _load_wav(path, header_only=True)
    return sr, wav[0], wav[1]


def load(path):
    path = str(path)
    if path.lower().endswith(".flac"):
        sr, ch, n = info(path)
        out = torch.empty(ch, n, dtype=torch.float32)
        ok = ctypes.c_int()
        _lib.check(_lib.load().cpc_flac_decode_f32(path.encode(), _lib.ptr(out), out.numel(), ctypes.byref(ok)),
                   f"flac_decode({path})")
        return out, sr
    return _load_wav(path)


def _load_wav(path, header_only=False):
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file (only .flac and .wav are supported)")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", body, 0)
        elif tag == b"data":
            payload = body
            break
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    code, channels, rate, _byte_rate, block_align, bits = fmt
    if code == 0xFFFE and len(data) > 0:          # WAVE_FORMAT_EXTENSIBLE: the real code follows
        code = 3 if bits == 32 and b"\x03\x00\x00\x00\x00\x00\x10\x00" in data[:128] else 1
    frames = len(payload) // block_align
    if header_only:
        return (channels, frames), rate
    if code == 1 and bits == 16:
        x = np.frombuffer(payload, dtype="<i2", count=frames * channels).astype(np.float32) / 32768.0
    elif code == 1 and bits == 8:
        x = (np.frombuffer(payload, dtype=np.uint8, count=frames * channels).astype(np.float32) - 128.0) / 128.0
    elif code == 1 and bits == 24:
        b = np.frombuffer(payload, dtype=np.uint8, count=frames * channels * 3).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
    elif code == 1 and bits == 32:
        x = np.frombuffer(payload, dtype="<i4", count=frames * channels).astype(np.float32) / 2147483648.0
    elif code == 3 and bits == 32:
        x = np.frombuffer(payload, dtype="<f4", count=frames * channels).astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAV encoding (format {code}, {bits} bits)")
    return torch.from_numpy(x.reshape(frames, channels).T.copy()), rate
