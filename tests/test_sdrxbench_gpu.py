"""GPU: sdrbench/sdrxbench -- the counterpart of the reference's sdrangelbench CLI (same options, test types and result line,
sdrbench/mainbench.cpp, parserbench.cpp) over the sdrx:: mirror classes -- gives, on sdrangelbench's own test data
(default-seeded std::mt19937 + libstdc++ distributions), the samples the CPU oracle gives (oracle/sdrbench_kat)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "sdrbench", "sdrxbench")
KAT = os.path.join(ROOT, "oracle", "sdrbench_kat")


@pytest.mark.skipif(not (os.path.exists(BENCH) and os.path.exists(KAT)), reason="sdrxbench / sdrbench_kat not built (python -c 'import __graft_entry__ as g; g.build()')")
@pytest.mark.parametrize("test,log2", [("decimateii", 4), ("decimateii", 6), ("decimateinfii", 5), ("decimatesupii", 3), ("decimateii", 0),
                                       ("decimatefi", 6), ("decimateff", 3), ("decimateif", 4)])
def test_same_samples_as_the_oracle_on_sdrbench_data(test, log2):
    out = subprocess.run([BENCH, "-t", test, "-l", str(log2), "-n", "1048576", "-r", "2", "--hash"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    want = subprocess.run([KAT, test, str(log2), "1048576", "2"], capture_output=True, text=True, timeout=300)     # two repetitions on one object: state carries
    assert want.returncode == 0
    got_hash = [l for l in out.stdout.splitlines() if l.startswith("hash:")]
    assert got_hash and got_hash[0].strip() == want.stdout.strip(), (got_hash, want.stdout)
    # the reference's result line (MainBench::printResults): "<prefix>: ran test in <ns> ns - sample rate: <kS/s> kS/s"
    assert re.search(r"MainBench::testDecimate(II|FI|FF|IF): ran test in \d+ ns - sample rate: [0-9.e+]+ kS/s", out.stdout), out.stdout
    assert "input resident in HBM" in out.stdout


def test_defaults_and_bad_values_like_the_reference_parser():
    if not os.path.exists(BENCH):
        pytest.skip("sdrxbench not built")
    out = subprocess.run([BENCH, "-n", "5", "-l", "9"], capture_output=True, text=True, timeout=300)     # both invalid: defaults 1048576 and 4
    assert out.returncode == 0 and "number of samples invalid" in out.stderr and "log2 factor invalid" in out.stderr
    assert "MainBench::testDecimateII" in out.stdout
