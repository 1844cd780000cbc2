import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


def pytest_terminal_summary(terminalreporter):
    """Every use of the ReLU-flip allowance (tests/test_gpu_parity.py assert_grad_close) is reported: which tensor, how many
    entries missed the 1e-4 bar, how many of those the fp64 oracle explains, the worst miss."""
    try:
        from tests.test_gpu_parity import FLIP_EVENTS, FORCED_EVENTS
    except Exception:
        return
    out = os.path.join(REPO, "gpurun_out")
    if FORCED_EVENTS:
        terminalreporter.write_sep("-", f"forced-oracle proofs: {len(FORCED_EVENTS)}")
        for e in FORCED_EVENTS:
            terminalreporter.write_line(f"forced-oracle {e['name']}: {e['forced_units']} units forced over {e['visited_samples']} samples "
                                        f"(largest forced margin {e.get('max_forced_margin', 0.0):.1e}, {e.get('near_tie_units')} near-tie units in the oracle); "
                                        f"worst {max(e['worst'].values()):.2e} ({max(e['worst'], key=e['worst'].get)})")
        if os.path.isdir(out):
            import json
            with open(os.path.join(out, "forced_oracle.json"), "w") as f:
                json.dump(FORCED_EVENTS, f, indent=1)
    if not FLIP_EVENTS:
        if FORCED_EVENTS and os.path.isdir(out):  # a GPU run that took no allowance says so in its record
            import json
            with open(os.path.join(out, "flip_allowance.json"), "w") as f:
                json.dump([], f)
        return
    terminalreporter.write_sep("-", f"ReLU-flip allowance taken {len(FLIP_EVENTS)} time(s)")
    for e in FLIP_EVENTS:
        terminalreporter.write_line(
            f"flip-allowance {e['name']}: {e['n_off']} entries above {e['tol']:g} ({e['explained']} agree with the second (fp64 / fp32) "
            f"oracle, {e['unexplained']} counted <= {e['allowed']} allowed, the worst of them {e['worst_unexplained']:.2e}), "
            f"worst {e['worst']:.2e}, rel L2 {e['l2']:.2e}, near-tie samples {e.get('near_tie_samples')}")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "flip_allowance.json"), "w") as f:
            json.dump(FLIP_EVENTS, f, indent=1)
