"""Child process of tests/test_abi_asan.py: random argument fuzz of EVERY C-ABI entry point of a host-only AddressSanitizer / UBSan build of
libjammy_hip.so (no device code, no GPU needed: argument validation and descriptor bookkeeping run on the host; whatever gets as far as a
launch fails cleanly with JF_ERR_LAUNCH).  Device pointers are never dereferenced by the host side, so arbitrary values stand in for them;
descriptor structs ARE read on the host and get random (including nonsensical) field values.  Prints `calls=<n> rc_hist=<...>`."""
import collections
import ctypes
import random
import sys

from jammy_flows_amd import _hip

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
lib = _hip.lib()
INTS = [0, 1, -1, 2, 3, 4, 7, 8, 9, 16, 17, 32, 33, 64, 128, 129, 548, 4096, 1 << 20, (1 << 31) - 1, -(1 << 31), 1 << 40, -(1 << 40)]
host_buf = (ctypes.c_double * 4096)()
PTRS = [None, 0x1000, 0x7F0000001000, ctypes.addressof(host_buf), ctypes.addressof(host_buf) + 4]


def rand_struct_array(ptr_type):
    cls = ptr_type._type_
    if rng.random() < 0.1:
        return None
    n = 9                                          # >= JF_MAX_CHAIN / JF_MAX_MCHAIN: the library may read as many as n_layers says, up to its maximum
    arr = (cls * n)()
    for i in range(n):
        for name, ftype in cls._fields_:
            if isinstance(getattr(arr[i], name), (int, float)):
                if ftype in (ctypes.c_double, ctypes.c_float):
                    setattr(arr[i], name, rng.choice([0.0, -1.0, 1e-3, 0.01, 1.0, 100.0, 1e300, float("inf"), float("nan")]))
                else:
                    v = rng.choice([0, 0, 1, 1, 2, 3, 4, 5, 8, 10, 11, 16, 17, 33, 64, -1, -2, -3, 1000, (1 << 31) - 1, -(1 << 31)])
                    try:
                        setattr(arr[i], name, v)
                    except (TypeError, ValueError):
                        pass
    return arr


def rand_arg(t):
    if t is _hip._P or t is ctypes.c_void_p:
        return rng.choice(PTRS)
    if t in (ctypes.c_int64, ctypes.c_int32, ctypes.c_int):
        v = rng.choice(INTS)
        if t is not ctypes.c_int64:
            v = max(-(1 << 31), min((1 << 31) - 1, v))
        return v
    if t is ctypes.c_double:
        return rng.choice([0.0, 1.0, -1.0, 1e300, float("nan")])
    if hasattr(t, "_type_") and not hasattr(t._type_, "_fields_"):      # POINTER(scalar): host arrays the step-plan entry points read / write
        return None if rng.random() < 0.3 else (t._type_ * 64)()
    if hasattr(t, "_type_"):                       # POINTER(struct)
        return rand_struct_array(t)
    raise TypeError(t)


funcs = []
for base, argtypes in _hip._SIGNATURES.items():
    for suf in ("_f32", "_f64"):
        funcs.append((base + suf, argtypes))
for name, (argtypes, _) in _hip._SIGNATURES_SINGLE.items():
    funcs.append((name, argtypes))
hist = collections.Counter()
keep = []
for i in range(n_calls):
    name, argtypes = funcs[i % len(funcs)] if i < 2 * len(funcs) else rng.choice(funcs)
    args = [rand_arg(t) for t in argtypes]
    keep = args                                    # struct arrays stay alive during the call
    rc = getattr(lib, name)(*args)
    hist[int(rc) if -10 < int(rc) < 1 else "value"] += 1
print("calls=%d functions=%d rc_hist=%s" % (n_calls, len(funcs), dict(hist)))
