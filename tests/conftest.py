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


@pytest.fixture(params=["persistent", "stepwise", "bulk"])
def decoder_path(request):
    """All three note decoders against the same reference numbers: the persistent few-clip decoder
    (csrc/a2s_dec_persist.hip, what a <= 8-clip fixture takes by default), the launch-per-step few-row kernels (dec_gru_step, dec_out_step,
    row_list tails: the long-clip chain of the 256-clip benchmark) and "bulk": the launch-per-step loop of the calls over hundreds of rows --
    library-style query / output products, attn_*_split256[_mq] and the round-6 mid-size fused kernels dec_gru_mid / dec_bwd_mid -- forced onto
    the small fixtures by switching the few-row path off.  Yields a checker: check(engine) asserts that the decoder calls of the engine's last
    forward really took that path (the library counts its persistent and mid-size launches)."""
    from piano_a2s_amd import hip
    L = hip.lib()
    on = request.param == "persistent"
    bulk = request.param == "bulk"
    _set_path(b"dec_persist", "A2S_DEC_PERSIST", on)
    _set_path(b"dec_fused", "A2S_DEC_FUSED", not bulk)
    start = L.a2s_debug_get(b"dec_persist_launches")
    start_mid = L.a2s_debug_get(b"dec_mid_launches")

    def check(eng):
        calls = [seg["staff"][k][2] for g in eng.saved["groups"] for seg in g["segments"] for k in ("up", "lo")]
        used = [sv.get("persist_ws") is not None for sv in calls]
        assert used and all(u == on for u in used), f"decoder path '{request.param}' was asked for, persistent launches prepared: {used}"
        # ... and the library really took it (it falls back to a launch per step when a precondition fails): its own count of persistent launches
        launches = L.a2s_debug_get(b"dec_persist_launches") - start
        assert (launches >= len(calls)) == on and (on or launches == 0), f"decoder path '{request.param}': {launches} persistent launches for {len(calls)} decoder calls"
        mids = L.a2s_debug_get(b"dec_mid_launches") - start_mid
        steps = sum(sv["launched"] for sv in calls)            # (forward steps enqueued: one dec_gru_mid each on the bulk path)
        assert (mids >= steps > 0) == bulk and (bulk or mids == 0), f"decoder path '{request.param}': {mids} mid-size launches for {steps} decoder steps"
    check.name = request.param
    yield check
    _set_path(b"dec_persist", "A2S_DEC_PERSIST", True)
    _set_path(b"dec_fused", "A2S_DEC_FUSED", True)


@pytest.fixture(params=["gru_persistent", "gru_stepwise"])
def encoder_path(request):
    """The encoder recurrences as one persistent launch per direction (csrc/a2s_persist.hip) or one launch per step (csrc/a2s_seq.hip)."""
    on = request.param == "gru_persistent"
    _set_path(b"gru_persist", "A2S_GRU_PERSIST", on)
    yield request.param
    _set_path(b"gru_persist", "A2S_GRU_PERSIST", True)
