import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def _set_path(key, env, on):
    from piano_a2s_amd import hip
    os.environ[env] = "1" if on else "0"
    hip.check(hip.lib().a2s_debug_set(key, 1 if on else 0), "a2s_debug_set")


@pytest.fixture(params=["persistent", "stepwise"])
def decoder_path(request):
    """Both note decoders against the same reference numbers: the persistent few-clip decoder (csrc/a2s_dec_persist.hip,
    what a <= 8-clip fixture takes by default) and the launch-per-step kernels the 256-clip benchmark times (attn_*_split256[_mq], dec_*_step,
    row_list tails).  Yields a checker: check(engine) asserts that the decoder calls of the engine's last forward really took that path."""
    from piano_a2s_amd import hip
    on = request.param == "persistent"
    _set_path(b"dec_persist", "A2S_DEC_PERSIST", on)
    start = hip.lib().a2s_debug_get(b"dec_persist_launches")

    def check(eng):
        calls = [seg["staff"][k][2] for g in eng.saved["groups"] for seg in g["segments"] for k in ("up", "lo")]
        used = [sv.get("persist_ws") is not None for sv in calls]
        assert used and all(u == on for u in used), f"decoder path '{request.param}' was asked for, persistent launches prepared: {used}"
        # ... and the library really took it (it falls back to a launch per step when a precondition fails): its own count of persistent launches
        launches = hip.lib().a2s_debug_get(b"dec_persist_launches") - start
        assert (launches >= len(calls)) == on and (on or launches == 0), f"decoder path '{request.param}': {launches} persistent launches for {len(calls)} decoder calls"
    check.name = request.param
    yield check
    _set_path(b"dec_persist", "A2S_DEC_PERSIST", True)


@pytest.fixture(params=["gru_persistent", "gru_stepwise"])
def encoder_path(request):
    """The encoder recurrences as one persistent launch per direction (csrc/a2s_persist.hip) or one launch per step (csrc/a2s_seq.hip)."""
    on = request.param == "gru_persistent"
    _set_path(b"gru_persist", "A2S_GRU_PERSIST", on)
    yield request.param
    _set_path(b"gru_persist", "A2S_GRU_PERSIST", True)
