"""Interpreter-mode stand-in for the reference's numpy_atomic (DISABLE_JIT behaviour,
/root/reference/pysco/numpy_atomic.py:59-67, 90-94): a plain read-modify-write."""


def atomic_add(ary, i, v):
    ary[i] += v


def atomic_sub(ary, i, v):
    ary[i] -= v


def atomic_max(ary, i, v):
    ary[i] = max(ary[i], v)


def atomic_min(ary, i, v):
    ary[i] = min(ary[i], v)
