"""Drop-in check at the reference's own C++ interfaces, on the GPU: oracle/_ref/dropin_test holds the
reference's DownChannelizer and Decimators/DecimatorsU classes (compiled from /root/reference in the build container,
where `make -C oracle dropin` / __graft_entry__.build() produce it) next to qt_adapter/GpuDownChannelizerBank and the
sdrx::Decimators mirror of include/sdrx/dsp.hpp, feeds both sides the same SampleVector spans / device-thread blocks
through BasebandSampleSink::feed() and decimateK_x(&it, buf, len), and compares every output sample and the
MsgChannelizerNotification each demod receives.  The binary needs Qt5Core of the image (/opt/conda/lib)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "dropin_test")
QT = "/opt/conda/lib/libQt5Core.so.5"

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not (os.path.exists(EXE) and os.path.exists(QT)), reason="drop-in harness not built (needs /root/reference + Qt at build time)")
def test_reference_objects_and_gpu_adapter_agree_in_one_process():
    env = dict(os.environ)
    sys_stdcpp = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"      # conda's lib dir (rpath for Qt) carries an older libstdc++
    if os.path.exists(sys_stdcpp):
        env["LD_PRELOAD"] = sys_stdcpp
    out = subprocess.run([EXE, "0"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, (out.returncode, out.stdout[-3000:], out.stderr[-2000:])
    assert "DROP-IN CHECK PASSED" in out.stdout
    assert out.stdout.count(" OK ") >= 12 * 2 + 12 + 6, out.stdout
    assert out.stdout.count("engine chain: decimate8_cen -> FIFO -> work() + DC corr -> channel") == 6
    assert "RefDec12: one object, K / fcPos changed at run time" in out.stdout
    assert "DecimatorsFI: one object, K / fcPos changed at run time" in out.stdout
    assert "MISMATCH" not in out.stdout


ENG = os.path.join(ROOT, "oracle", "_ref", "dropin_engine_test")


@pytest.mark.skipif(not (os.path.exists(ENG) and os.path.exists(QT)), reason="engine drop-in harness not built (needs /root/reference + Qt at build time)")
def test_real_engine_and_gpu_engine_agree_in_one_process():
    """oracle/dropin_engine_test.cpp: the REAL DSPDeviceSourceEngine (QThread + SyncMessenger, compiled from the reference's
    own dspdevicesourceengine.cpp) next to qt_adapter/GpuDeviceSourceEngine: same source FIFO traffic, same sinks; the sample
    streams the sinks receive (no correction / DC / DC + I/Q imbalance, reconfigured mid-stream), the channel outputs
    (3 real DownChannelizers vs one GpuDownChannelizerBank) and the state machine must agree."""
    env = dict(os.environ)
    sys_stdcpp = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    if os.path.exists(sys_stdcpp):
        env["LD_PRELOAD"] = sys_stdcpp
    out = subprocess.run([ENG], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, (out.returncode, out.stdout[-3000:], out.stderr[-2000:])
    assert "ENGINE DROP-IN: ALL OK" in out.stdout
    assert out.stdout.count("identical") == 3 + 3 + 1 and "DIFFERENT" not in out.stdout and "FAIL" not in out.stdout


def test_reference_sdrangelbench_sources_run_on_the_gpu_classes():
    """oracle/_ref/sdrangelbench_gpu = the reference's OWN sdrbench/mainbench.cpp + parserbench.cpp, compiled unchanged against
    qt_adapter/shadow/ (its Decimators / DecimatorsIF / FI / FF members are sdrx:: classes): every test type runs and prints
    the reference's result line.  (The samples themselves are checked by tests/test_sdrxbench_gpu.py on the same data.)"""
    exe = os.path.join(ROOT, "oracle", "_ref", "sdrangelbench_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/sdrangelbench_gpu not built (make -C oracle dropin_bench, build container only)")
    env = dict(os.environ)
    sys_stdcpp = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    if os.path.exists(sys_stdcpp):
        env["LD_PRELOAD"] = sys_stdcpp
    for test, prefix in (("decimateii", "testDecimateII"), ("decimateinfii", "testDecimateII"), ("decimatesupii", "testDecimateII"),
                         ("decimatefi", "testDecimateFI"), ("decimateff", "testDecimateFF"), ("decimateif", "testDecimateIF")):
        for log2 in (0, 4, 6):
            out = subprocess.run([exe, "-t", test, "-l", str(log2), "-n", "2000000", "-r", "3"], capture_output=True, text=True, timeout=300, env=env)
            text = out.stdout + out.stderr
            assert out.returncode == 0, (test, log2, text[-2000:])
            assert f"MainBench::{prefix}: ran test in" in text and "kS/s" in text, (test, log2, text[-1000:])
            assert "sdrx" not in text.lower() or "error" not in text.lower(), text[-1000:]


@pytest.mark.parametrize("log2,fcpos,bits", [(4, 2, 1), (6, 0, 1), (3, 1, 2), (5, 2, 0), (1, 0, 2), (0, 2, 1)])
def test_reference_device_thread_unchanged_on_gpu_decimators(log2, fcpos, bits, tmp_path):
    """plugins/samplesource/testsource/testsourcethread.cpp, compiled unchanged twice (oracle/dropin_thread_test.cpp): against the
    reference's Decimators and against qt_adapter/shadow (GPU classes).  Both feed the reference's SampleSinkFifo from a timer; the
    stream does not depend on the timer (2.56 MS/s: whole decimation groups per elapsed ms), only its length does -- the common
    prefix must be identical, sample for sample."""
    ref = os.path.join(ROOT, "oracle", "_ref", "testsource_ref"); gpu = os.path.join(ROOT, "oracle", "_ref", "testsource_gpu")
    if not (os.path.exists(ref) and os.path.exists(gpu)):
        pytest.skip("oracle/_ref/testsource_{ref,gpu} not built (make -C oracle dropin_thread, build container only)")
    env = dict(os.environ)
    sys_stdcpp = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    if os.path.exists(sys_stdcpp):
        env["LD_PRELOAD"] = sys_stdcpp
    outs = []
    for exe, name in ((ref, "ref.bin"), (gpu, "gpu.bin")):
        path = str(tmp_path / name)
        r = subprocess.run([exe, str(log2), str(fcpos), str(bits), path, "14"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, (exe, r.stdout[-500:], r.stderr[-1500:])
        outs.append(open(path, "rb").read())
    n = min(len(outs[0]), len(outs[1]))
    assert n >= 4 * 1000, (len(outs[0]), len(outs[1]))
    assert outs[0][:n] == outs[1][:n], (log2, fcpos, bits, n // 4)
