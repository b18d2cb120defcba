"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-buildable parts (no GPU sanitizers on this pool): the oracle
(oracle/fxref_*.c: every modulation / code / CRC, odd chunkings, resets mid-frame, the bare detector) and the product's
host-side generator (csrc/fx_codec.hpp, what flexframegen_* run on), each built with -fsanitize=address,undefined
-fno-sanitize-recover and run as a small driver (tests/cpp/oracle_sanitize.c, tests/cpp/codec_sanitize.cpp)."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def test_oracle_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_sanitize")
    src = sorted(glob.glob(os.path.join(ROOT, "oracle", "fxref_*.c")))
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-mfma", "-mavx2", "-ffp-contract=off", "-fno-fast-math", "-Wall"] + SAN +
                          ["-o", exe, os.path.join(ROOT, "tests", "cpp", "oracle_sanitize.c")] + src + ["-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "0 failures" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_host_frame_generator_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "codec_sanitize")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"] + SAN +
                          ["-o", exe, os.path.join(ROOT, "tests", "cpp", "codec_sanitize.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "frames" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
