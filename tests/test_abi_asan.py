"""Host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only -- GPU sanitizers are not available on
the pool): libjammy_hip.so is rebuilt with the sanitizers on its HOST side (`hipcc -fsanitize=address,undefined -fno-gpu-sanitize`, device code
at -O1) and every entry point is called a few thousand times with random pointers / sizes / strides / descriptor fields (tests/abi_fuzz_child.py).  A sanitizer report
or a crash fails the test; every call must come back with a status code."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _asan_runtime():
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return hits[0] if hits else None


@pytest.fixture(scope="module")
def asan_lib(tmp_path_factory):
    if not os.path.exists(HIPCC) or _asan_runtime() is None:
        pytest.skip("hipcc or its AddressSanitizer runtime is not available")
    out = tmp_path_factory.mktemp("asan")
    srcs = sorted(glob.glob(os.path.join(ROOT, "jammy_flows_amd", "csrc", "*.hip")))
    flags = ["--offload-arch=gfx950", "-O1", "-g0", "-std=c++20", "-fPIC", "-fsanitize=address,undefined", "-fno-gpu-sanitize",
             "-fno-sanitize-recover=undefined", "-Wno-unused-function", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
    import concurrent.futures

    def compile_one(src):
        obj = os.path.join(out, os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([HIPCC] + flags + ["-c", src, "-o", obj], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, srcs))
    lib = os.path.join(out, "libjammy_hip_asan.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-fno-gpu-sanitize"] + objs + ["-o", lib],
                          stderr=subprocess.DEVNULL)
    return lib


@pytest.mark.parametrize("seed", [1, 2])
def test_every_entry_point_survives_random_arguments_under_asan_ubsan(asan_lib, seed):
    env = dict(os.environ, JF_LIB_PATH=asan_lib, LD_PRELOAD=_asan_runtime(), PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_fuzz_child.py"), str(seed), "3000"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-4000:])
    assert "calls=3000" in r.stdout, r.stdout
