"""Untrusted bytes: sanitizer build + fuzzing of the bitstream parsers (SURVEY section 5 "sanitizers").

The reference reads its files back with no validation at all (`model/model.py:314-385` load_bitstream, `:443-486` the tmc3
round trip).  Here every parser of file bytes must either decode or raise `PccError` / return `PCC_EINVAL`:
  * `tests/fuzz/fuzz_host.cpp`, built with -fsanitize=address,undefined (host side, CPU only): the octree coder of the latent
    coordinates and the single-stream rANS decoder on >= 10 000 seeded mutations (truncated, bit-flipped, length-lying, padded,
    noise) + clean round trips -- a sanitizer report aborts the run;
  * `container.load_bitstream` / `decode_points` (Python on the same host functions): seeded mutations of real files.
"""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc for the sanitizer build")
def test_sanitizer_build_of_the_host_parsers_survives_60k_mutated_streams():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "fuzz")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    exe = os.path.join(ROOT, "tests", "fuzz", "_build", "fuzz_host")
    total = 0
    for seed in range(1, 6):
        r = subprocess.run([exe, "12000", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1"))
        assert r.returncode == 0 and "FUZZ OK" in r.stdout, r.stdout[-3000:]
        total += int(r.stdout.split("FUZZ OK:")[1].split()[0])
    assert total >= 10000


def _cells(seed, depth, n):
    rng = np.random.default_rng(seed)
    c = np.unique(rng.integers(0, 1 << depth, (n, 3)), axis=0)
    return c.astype(np.int64)


def _mutations(rng, data, count):
    for _ in range(count):
        b = bytearray(data)
        kind = int(rng.integers(0, 6))
        if kind == 0:
            b = b[:int(rng.integers(0, len(b) + 1))]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 9))):
                if b:
                    b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2 and len(b) >= 16:
            off = int(rng.integers(0, (len(b) - 4) // 4)) * 4              # an aligned int32 field made to lie
            struct.pack_into("<i", b, off, int(rng.choice([-1, 0, 1, 2 ** 31 - 1, -2 ** 31, int(rng.integers(0, 1 << 24))])))
        elif kind == 3:
            b += bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
        elif kind == 4:
            for i in range(len(b)):
                if rng.integers(0, 16) == 0:
                    b[i] = int(rng.integers(0, 256))
        else:
            b = bytearray(rng.integers(0, 256, int(rng.integers(0, 96)), dtype=np.uint8).tobytes())
        yield bytes(b)


def test_container_parser_rejects_or_decodes_every_mutated_file(tmp_path):
    """2 000 seeded mutations of two real container files (one and three blocks) through `load_bitstream`, 1 000 of a
    latent-coordinate stream through `decode_points`: PccError or a structurally valid result, never another exception, never
    an allocation sized by a lying header (the point count is bounded before anything is allocated for it)."""
    from unified_point_cloud_compression_amd import container, lib as L
    rng = np.random.default_rng(7)
    files = []
    for nblocks in (1, 3):
        coords, strings, shapes, ks, qs = [], [], [], [], []
        for b in range(nblocks):
            c = _cells(10 * nblocks + b, 5, 150) * 8
            coords.append(torch.from_numpy(np.concatenate([np.zeros((len(c), 1), np.int64), c], 1)).int())
            y = container.StreamBytes(bytes(rng.integers(0, 256, 200, dtype=np.uint8)), 16)
            z = container.StreamBytes(bytes(rng.integers(0, 256, 60, dtype=np.uint8)), 4)
            strings.append([[y], [z]])
            shapes.append([17])
            ks.append([[5], [50], [500]])
            qs.append(torch.tensor([[0.5, 0.25]]))
        path = os.path.join(tmp_path, f"ok{nblocks}.bin")
        container.save_bitstream(path, coords, strings, shapes, ks, qs)
        out = container.load_bitstream(path)
        assert len(out[0]) == nblocks
        files.append(open(path, "rb").read())
    decoded = rejected = 0
    p = os.path.join(tmp_path, "m.bin")
    for data in files:
        for bad in _mutations(rng, data, 1000):
            open(p, "wb").write(bad)
            try:
                c, s, sh, k, q = container.load_bitstream(p)
                assert len(c) == len(s) == len(sh) == len(k) == len(q)
                assert all(x.dim() == 2 and x.shape[1] == 3 and x.shape[0] <= 4096 * len(bad) + 4096 for x in c)
                decoded += 1
            except L.PccError:
                rejected += 1
    pts = container.encode_points(_cells(3, 6, 900) * 8)
    for bad in _mutations(rng, pts, 1000):
        try:
            xyz = container.decode_points(bad)
            assert xyz.ndim == 2 and xyz.shape[1] == 3
            decoded += 1
        except L.PccError:
            rejected += 1
    assert decoded + rejected == 3000 and rejected > 1000 and decoded > 100, (decoded, rejected)
