"""GPU parity (-m gpu): the 16-byte pair-row output streams of the maxiSample / maxiEnvGen kernels (round 4; knob rw_store, emit_chunk
in csrc/mxg_common.h).  A store flavour must not change a bit, so the parity tests of those families are simply run again under every
setting of the knob (2 / 3 / 4: pair rows with plain / write-through / non-temporal stores; their own runs cover the automatic rule,
which keeps small blocks on the 8-byte stores)."""
import pytest

import test_gpu_envgen as EG
import test_gpu_sample as SM

pytestmark = pytest.mark.gpu

RW = [2, 3, 4]


class _Knob:
    def __init__(self, mx, value):
        self.L, self.value = mx.lib(), value

    def __enter__(self):
        self.prev = self.L.mxg_tune(b"rw_store", self.value)

    def __exit__(self, *a):
        self.L.mxg_tune(b"rw_store", self.prev)


@pytest.mark.parametrize("rw", RW)
@pytest.mark.parametrize("mode", range(9))
def test_sample_players_golden_and_ragged(mx, golden, port, rw, mode):
    with _Knob(mx, rw):
        SM.test_sample_golden(mx, golden, mode)
        for N in (7, 8, 21, 203):
            SM.test_sample_ragged_blocks(mx, golden, port, mode, N)


@pytest.mark.parametrize("rw", RW)
def test_sample_per_sample_speed_and_triggers(mx, golden, port, rw):
    with _Knob(mx, rw):
        for mode in (4, 5, 6, 7, 8):
            SM.test_sample_per_sample_speed_ragged(mx, golden, port, mode, 203)
        for mode in range(5):
            SM.test_sample_on_zx(mx, port, mode, 203)
        SM.test_sample_on_zx_per_sample_speed(mx, port)
        SM.test_sample_play_with_phasor(mx, port, 301)
        SM.test_sample_vs_oracle_large(mx, port)


@pytest.mark.parametrize("rw", RW)
@pytest.mark.parametrize("mode", [4, 5, 6, 7, 8])
def test_sample_speed_players_time_parts(mx, port, rw, mode):
    with _Knob(mx, rw):
        for Ls, split in [(20000, 0), (700, 3), (20000, 8), (700, -3)]:
            SM.test_sample_speed_players_full_waves(mx, port, mode, Ls, split)
        if mode <= 6:
            SM.test_sample_time_parts_corner_heads(mx, port, mode)


@pytest.mark.parametrize("rw", RW)
def test_envgen(mx, port, rw):
    with _Knob(mx, rw):
        for name in EG.CASES:
            EG.test_envgen_per_voice_triggers(mx, port, name, 1, 1)
            EG.test_envgen_per_voice_triggers(mx, port, name, 0, 0)
        EG.test_envgen_shared_gate_and_helpers(mx, port)
        for shape in ("AR", "ADSR"):
            EG.test_envgen_shared_gate_steady_states(mx, port, shape)
        EG.test_envgen_uploaded_state_parked_on_the_end_test(mx, port, 0)
        EG.test_envgen_uploaded_state_parked_on_the_end_test(mx, port, 1)
