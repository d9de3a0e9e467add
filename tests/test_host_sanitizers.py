"""CPU: the host half of the library (alist reader, code constructions, encoders, parsers) built
with AddressSanitizer + UndefinedBehaviorSanitizer and driven over valid and malformed inputs.
(GPU sanitizers are not available on the pool; the device code is covered by the parity tests.)"""
import os
import subprocess

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "ldpc_toolbox_amd", "csrc")


def test_host_code_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitizer_driver")
    srcs = [os.path.join(ROOT, "tests", "host_sanitizer_driver.cpp")] + \
           [os.path.join(CSRC, f) for f in ("sparse.cpp", "codes.cpp", "encoder.cpp", "implementation.cpp")]
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-fno-omit-frame-pointer", "-o", exe] + srcs, check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host sanitizer driver: ok" in r.stdout


def test_oracle_is_clean_under_asan_ubsan(tmp_path):
    """the checker itself: every implementation name on a small 5G NR graph (rows of degree 3..19)"""
    import ldpc_toolbox_amd as lt
    exe = str(tmp_path / "oracle_sanitizer_driver")
    subprocess.run(["gcc", "-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-fno-omit-frame-pointer", "-pthread", "-o", exe,
                    os.path.join(ROOT, "tests", "oracle_sanitizer_driver.c"), os.path.join(ROOT, "oracle", "ldpc_oracle.c"),
                    "-lm"], check=True, capture_output=True)
    alist = tmp_path / "h.alist"
    alist.write_text(lt.code_alist("nr5g:1:4"))
    r = subprocess.run([exe, str(alist)] + list(lt.ALL_IMPLEMENTATIONS), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "oracle sanitizer driver: ok" in r.stdout


def test_layered_schedule_tables_under_asan_ubsan(tmp_path):
    """dependency levels and the task records of the slice-persistent layered kernel (csrc/slice_tasks.h): every row in
    exactly one task of its level, every edge in exactly one lane slot (the middle edge of an odd shared row in two),
    out-of-range padding everywhere else -- for codes with short rows only, with rows of 19 (5G NR BG1: shared by two
    lanes), with rows too long for any task (DVB-S2 short 8/9: the table says so), and a staircase code"""
    import ldpc_toolbox_amd as lt
    exe = str(tmp_path / "slice_tasks_driver")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-fno-omit-frame-pointer", "-o", exe, os.path.join(ROOT, "tests", "slice_tasks_driver.cpp"),
                    os.path.join(CSRC, "sparse.cpp")], check=True, capture_output=True)
    files = []
    for spec in ("nr5g:1:8", "nr5g:2:24", "ar4ja:1/2:1024", "dvbs2:R8_9short", "nr5g:1:384"):
        f = tmp_path / (spec.replace(":", "_").replace("/", "_") + ".alist")
        f.write_text(lt.code_alist(spec))
        files.append(str(f))
    r = subprocess.run([exe] + files, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "slice tasks driver: ok" in r.stdout


def test_host_threads_are_clean_under_tsan(tmp_path):
    """SURVEY section 5 "Race detection": the library's real host code (c_api.cpp, device_decoder.hip, simulator.hip
    compiled host-only) under ThreadSanitizer against tests/hip_stub -- streams are worker threads, events are counters,
    kernels publish the progress word.  Scenarios: every group runs all its iterations; the device "finishes" at
    iteration 3 / 1 (the enqueuing threads' early exits and pacing); a HIP call fails somewhere in the middle (the
    error returns of decode_host / decode_device with lane threads alive).  Round 4 found and fixed two unsynchronised
    writes this way (skew_record_ and last_persist_ written by both lanes' threads)."""
    out = str(tmp_path / "tsan")
    b = subprocess.run([os.path.join(ROOT, "tests", "hip_stub", "build.sh"), out], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-4000:]
    exe = os.path.join(out, "tsan_driver")
    scenarios = [({}, "0"), ({"HIP_STUB_DONE_AT": "3"}, "0"), ({"HIP_STUB_DONE_AT": "1"}, "0")]
    scenarios += [({"HIP_STUB_FAIL": f}, "1") for f in ("hipEventRecord:3", "hipEventRecord:40", "hipStreamWaitEvent:2",
                                                        "hipStreamWaitEvent:25", "hipLaunchKernel:7000", "hipMemcpyAsync:9",
                                                        "hipMalloc:12", "hipStreamSynchronize:6")]
    for extra, expect_error in scenarios:
        env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66", **extra)
        r = subprocess.run([exe, expect_error], capture_output=True, text=True, env=env, timeout=600)
        text = r.stdout + r.stderr
        assert "ThreadSanitizer" not in text, (extra, text[-6000:])
        assert r.returncode == 0 and "tsan driver: ok" in r.stdout, (extra, text[-3000:])
