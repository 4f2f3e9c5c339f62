"""Helpers shared by the -m gpu parity tests."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, dtype=np.complex64).view(np.uint64)


def assert_bit_exact(got, ref, what=""):
    got = np.ascontiguousarray(got, dtype=np.complex64)
    ref = np.ascontiguousarray(ref, dtype=np.complex64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    bad = np.nonzero(bits(got).ravel() != bits(ref).ravel())[0]
    if bad.size:
        i = int(bad[0])
        raise AssertionError("%s: %d of %d outputs differ, first at %d: got %r want %r" % (
            what, bad.size, got.size, i, got.ravel()[i], ref.ravel()[i]))


def ulp_distance(a, b):
    a = np.ascontiguousarray(a, dtype=np.complex64).view(np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.complex64).view(np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def to_gpu(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def to_cpu(t):
    return t.cpu().numpy()
