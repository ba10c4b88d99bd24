"""GPU (-m gpu): the per-voice wavetable EXTENSION (mxg_osc_render_tables, csrc/osctab.hip; SURVEY 8(d) row 2's optional HBM-read
variant -- the reference has one shared sineBuffer, C:63).  Parity: (a) every table = sineBuffer  =>  the bits of the shared-table
sinebuf, i.e. the reference's; (b) arbitrary tables against the port's restatement of C:266-274 with a table argument
(oracle/maxi_oracle.c mxo_osc_tables, itself pinned by (a) on the CPU); the fused mixdown within the mix tolerance."""
import numpy as np
import pytest

from conftest import assert_bits_equal, mix_tol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V,N", [(1000, 512), (64, 512), (4099, 300), (16, 37), (17, 16), (5000, 256), (40000, 512)])
def test_tables_bank_with_the_shared_table_is_sinebuf(mx, port, V, N):
    rng = np.random.default_rng(V + N)
    freq = rng.uniform(20, 20000, V)
    tabs = np.tile(port.sine_table(), (V, 1))
    bank = mx.maxiOscBank(V)
    ref = mx.maxiOscBank(V)
    d_tabs = mx.DeviceBuffer.from_numpy(tabs)
    for blk in range(2):
        got, _ = bank.sinebuf_tables(freq, d_tabs, N)
        exp = ref.sinebuf(freq, N)
        assert_bits_equal(got.numpy(), exp.numpy(), "block %d" % blk)
    assert_bits_equal(bank.phase.numpy(), ref.phase.numpy(), "phase")
    assert_bits_equal(bank.output.numpy(), ref.output.numpy(), "output member")


@pytest.mark.parametrize("V,N", [(1000, 512), (300, 100), (4099, 333), (33, 512)])
def test_tables_bank_against_the_oracle(mx, port, V, N):
    rng = np.random.default_rng(3 * V + N)
    freq = rng.uniform(20, 20000, V)
    pan = rng.uniform(-0.1, 1.1, V)
    tabs = rng.uniform(-1, 1, (V, 514))
    bank = mx.maxiOscBank(V)
    d_tabs = mx.DeviceBuffer.from_numpy(tabs)
    o1, m1 = bank.sinebuf_tables(freq, d_tabs, N, pan=pan)
    o2, m2 = bank.sinebuf_tables(freq, d_tabs, N, pan=pan)
    eo, eph, ehd = port.osc_tables(freq, tabs, 2 * N)
    assert_bits_equal(np.concatenate([o1.numpy(), o2.numpy()]), eo, "per-voice tables")
    assert_bits_equal(bank.phase.numpy(), eph, "phase")
    assert_bits_equal(bank.output.numpy(), ehd, "output member")
    em = port.mix_stereo(eo, pan)
    m = np.concatenate([m1.numpy(), m2.numpy()])
    assert np.abs(m - em).max() <= mix_tol(V, np.abs(eo).max())
    # mix only (no per-voice block: the measured form): the same mix bits, the same carried state
    bank2 = mx.maxiOscBank(V)
    none, m3 = bank2.sinebuf_tables(freq, d_tabs, N, pan=pan, store=False)
    assert none is None
    assert_bits_equal(m3.numpy(), m1.numpy(), "mix-only form")
    none, m4 = bank2.sinebuf_tables(freq, d_tabs, N, pan=pan, store=False)
    assert_bits_equal(m4.numpy(), m2.numpy(), "mix-only form, second block")
    assert_bits_equal(bank2.phase.numpy(), eph, "phase, mix-only form")


@pytest.mark.parametrize("V,N", [(1000, 512), (300, 100), (4099, 333), (33, 512), (17, 16), (40000, 512), (131072, 512), (131073, 512)])
@pytest.mark.parametrize("store", [False, True])
def test_tables_pipelined_blocks_and_fused_row_sum(mx, port, V, N, store):
    """Round 6: mxg_osc_render_tables_ex with MXG_TABLES_AHEAD -- the render of block k also walks block k + 1's
    phase recurrence, so from the second call on a call is one kernel -- and with d_mix, the row sum formed inside the render kernel by
    the workgroups that finish last.  Five carried blocks: every block, every mix, the carried phase after EVERY call and the `output`
    member must be the plain two-kernel path's bits (mxg_osc_render_tables + mxg_mix_rows_sum); then the pattern is broken (a call
    without the flag, a call with another N) and picked up again, still the same bits.  131 073 voices: more than 512 voices per
    workgroup, the flag is ignored."""
    if V > 100000 and store:
        pytest.skip("per-voice blocks of the large banks are covered by the mix-only form")
    L = mx.lib()
    # (the row sum inside the kernel is written for the side-by-side form, knob tab_sides 2: the reference path takes the same decomposition
    # of the bank into workgroup rows, so that the two mixes are the same additions; the one-round form is compared below)
    prev_sides = L.mxg_tune(b"tab_sides", 2)
    try:
        _pipelined_blocks(mx, port, V, N, store, fused_sum=True)
    finally:
        L.mxg_tune(b"tab_sides", prev_sides)
    _pipelined_blocks(mx, port, V, N, store, fused_sum=False)


def _pipelined_blocks(mx, port, V, N, store, fused_sum):
    rng = np.random.default_rng(5 * V + N)
    freq = mx.DeviceBuffer.from_numpy(rng.uniform(20, 20000, V))
    pan = mx.DeviceBuffer.from_numpy(rng.uniform(-0.1, 1.1, V))
    d_tabs = mx.DeviceBuffer.from_numpy(rng.uniform(-1, 1, (V, 514)))
    ref, bank = mx.maxiOscBank(V), mx.maxiOscBank(V)
    plan = [(N, True), (N, True), (N, True), (N, False), (N, True), (max(N // 2, 1), True), (N, True), (N, True)]
    for k, (n, ahead) in enumerate(plan):
        eo, em = ref.sinebuf_tables(freq, d_tabs, n, pan=pan, store=store)
        go, gm = bank.sinebuf_tables(freq, d_tabs, n, pan=pan, store=store, ahead=ahead, fused_sum=fused_sum)
        if store:
            assert_bits_equal(go.numpy(), eo.numpy(), "block %d" % k)
        assert_bits_equal(gm.numpy(), em.numpy(), "mix %d (%s)" % (k, "row sum inside the kernel" if fused_sum else "pipelined, row-sum kernel"))
        assert_bits_equal(bank.phase.numpy(), ref.phase.numpy(), "d_phase after call %d" % k)
        assert_bits_equal(bank.output.numpy(), ref.output.numpy(), "output member after call %d" % k)
    # and against the oracle (small banks: the port walks every sample)
    if V <= 5000:
        fh, th = freq.numpy(), d_tabs.numpy()
        total = sum(n for n, _ in plan)
        eo, eph, ehd = port.osc_tables(fh, th, total)
        assert_bits_equal(bank.phase.numpy(), eph, "phase vs the oracle")
        assert_bits_equal(bank.output.numpy(), ehd, "output member vs the oracle")


@pytest.mark.parametrize("V,N", [(1000, 512), (4099, 333), (17, 16), (40000, 512), (131072, 512)])
def test_tables_workgroup_forms_agree(mx, port, V, N):
    """Knob tab_sides: 1 = workgroups of 256 lanes, one round of eight voices at a time, two per CU (the default since round 6); 2 =
    workgroups of 512 lanes, two rounds side by side (rounds 4-5).  The per-voice block, the carried phase and the `output` member are the
    same bits; the mix is the same products added in another decomposition of the bank (one row per workgroup): within mix_tol of each
    other and of the oracle's sequential sum."""
    rng = np.random.default_rng(7 * V + N)
    freq, pan = rng.uniform(20, 20000, V), rng.uniform(-0.1, 1.1, V)
    tabs = rng.uniform(-1, 1, (V, 514))
    d_tabs = mx.DeviceBuffer.from_numpy(tabs)
    L = mx.lib()
    res = {}
    for sides in (1, 2):
        prev = L.mxg_tune(b"tab_sides", sides)
        try:
            bank = mx.maxiOscBank(V)
            o, m = [], []
            for _ in range(2):
                a, b = bank.sinebuf_tables(freq, d_tabs, N, pan=pan, store=V <= 40000)
                o.append(None if a is None else a.numpy()); m.append(b.numpy())
            res[sides] = (o, m, bank.phase.numpy(), bank.output.numpy())
        finally:
            L.mxg_tune(b"tab_sides", prev)
    for k in range(2):
        if res[1][0][k] is not None:
            assert_bits_equal(res[1][0][k], res[2][0][k], "block %d" % k)
        assert np.abs(res[1][1][k] - res[2][1][k]).max() <= mix_tol(V, 1.0, sums=res[2][1][k])
    assert_bits_equal(res[1][2], res[2][2], "phase")
    assert_bits_equal(res[1][3], res[2][3], "output member")
    if V <= 5000:
        eo, eph, ehd = port.osc_tables(freq, tabs, 2 * N)
        assert_bits_equal(np.concatenate(res[1][0]), eo, "one-round form vs the oracle")
        assert_bits_equal(res[1][2], eph, "phase vs the oracle")
        em = port.mix_stereo(eo, pan)
        assert np.abs(np.concatenate(res[1][1]) - em).max() <= mix_tol(V, np.abs(eo).max())


def test_tables_invalid(mx):
    L = mx.lib()
    V = 32
    bank = mx.maxiOscBank(V)
    f = mx.DeviceBuffer.from_numpy(np.full(V, 100.0))
    t = mx.DeviceBuffer((V, 514))
    out = mx.DeviceBuffer((600, V))
    assert L.mxg_osc_render_tables(V, 600, f.ptr, t.ptr, bank.phase.ptr, bank.output.ptr, out.ptr, None, None, None) == -1   # N > 512
    assert b"512" in L.mxg_last_error()
    assert L.mxg_osc_render_tables(V, 16, f.ptr, t.ptr, bank.phase.ptr, bank.output.ptr, None, None, None, None) == -1     # nothing to produce
    assert L.mxg_osc_render_tables(V, 16, f.ptr, None, bank.phase.ptr, bank.output.ptr, out.ptr, None, None, None) == -1
    assert L.mxg_osc_render_tables(0, 16, f.ptr, t.ptr, bank.phase.ptr, bank.output.ptr, out.ptr, None, None, None) == 0
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count  # (256 on an unpartitioned MI355X; ADVICE r04: do not hard-code it)
    assert L.mxg_osc_tables_groups(1) == 1 and L.mxg_osc_tables_groups(1 << 20) == min((1 << 20) // 8, 2 * cus)  # (tab_sides 1: two workgroups per CU)
