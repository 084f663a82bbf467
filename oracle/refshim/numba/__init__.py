"""No-op stand-in for `numba`, used ONLY by oracle/gen_golden.py inside the build container.

The reference's kernels are plain Python functions decorated with @njit; with the decorator
turned into the identity (the reference's own "DISABLE_JIT" mode, numpy_atomic.py:9-10) they
execute under the interpreter.  Nothing here ships in the product or travels to the GPU box
as anything but a test-infrastructure file; the golden fixtures under tests/golden/ are the
only artefacts kept from such a run.
"""
prange = range


class _Config:
    DISABLE_JIT = True


config = _Config()
_nthreads = 1


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(f):
        return f

    return deco


jit = njit


def set_num_threads(n):
    global _nthreads
    _nthreads = int(n)


def get_num_threads():
    return _nthreads
