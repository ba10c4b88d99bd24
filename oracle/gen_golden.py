#!/usr/bin/env python3
"""oracle/gen_golden.py -- generate tests/golden/*.npz from the REFERENCE itself.

Runs the unmodified reference (oracle/_ref/libmaxiref.so = /root/reference sources +
oracle/ref_harness.cpp, g++ -O2 -ffp-contract=off) on small seeded inputs and stores inputs,
outputs and final state.  The reference ships no golden vectors of its own, so
these fixtures are what pins both the plain-C oracle and the HIP path.  Only runnable in the
build container (needs /root/reference); the .npz files and this script are committed.

    python oracle/gen_golden.py              # rewrites tests/golden/*.npz + MANIFEST.json
    python oracle/gen_golden.py extra.npz    # rewrites only the named files (+ the manifest)
"""
import hashlib
import json
import os
import platform
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF_SRC = "/root/reference/src"
SEED = 0x4D415849  # "MAXI" 


ONLY = set(sys.argv[1:])


def save(fname, **arrays):
    """np.savez_compressed into tests/golden, unless a selection was given and fname is not in it."""
    if ONLY and fname not in ONLY:
        return
    np.savez_compressed(os.path.join(GOLD, fname), **arrays)


def osc_inputs(V):
    rng = np.random.default_rng(SEED)
    freq = np.concatenate([[0.2, 20.0, 55.0, 440.0, 86.1328125, 43.06640625, 11025.0, 22050.0],
                           rng.uniform(0.1, 20000.0, V - 8)])
    p2 = rng.uniform(0.3, 1.0, V)
    p1 = p2 * rng.uniform(0.0, 0.9, V)
    duty = rng.uniform(-0.1, 1.1, V)
    return freq, p1, p2, duty


def main():
    os.makedirs(GOLD, exist_ok=True)
    R = pyoracle.reference()
    assert R.kind == "reference"
    R.settings(44100, 2, 1024)
    files = {}

    # ---- maxiOsc: every waveform, 2 consecutive blocks (state carry) ----------------------
    V, N = 32, 160
    freq, p1, p2, duty = osc_inputs(V)
    names = ["sinewave", "coswave", "phasor", "saw", "triangle", "square", "pulse", "impulse",
             "sinebuf", "sinebuf4", "sawn", "phasorBetween"]
    d = dict(freq=freq, p1=p1, p2=p2, duty=duty, N=N)
    for wf, name in enumerate(names):
        a, b = (duty, None) if name == "pulse" else (p1, p2)
        o1, ph, hd = R.osc(wf, freq, N, p1=a, p2=b)
        o2, ph2, hd2 = R.osc(wf, freq, N, phase=ph, hold=hd, p1=a, p2=b)
        d["out_" + name] = np.concatenate([o1, o2])
        d["phase_" + name] = ph2
        d["hold_" + name] = hd2
    # audio-rate frequency modulation (fps=1) on sinebuf and saw
    rng = np.random.default_rng(SEED + 1)
    fm = freq[None, :] * (1.0 + 0.25 * rng.uniform(-1, 1, (N, V)))
    d["fm"] = fm
    for name in ("sinebuf", "saw", "sawn"):
        o, ph, hd = R.osc(names.index(name), fm, N, per_sample=True)
        d["fm_out_" + name] = o
        d["fm_phase_" + name] = ph
    # known answer recorded in SURVEY.md 8a (a2): sinewave(440), samples 0..3
    ka, _, _ = R.osc(0, np.array([440.0]), 4)
    assert [repr(float(x)) for x in ka[:, 0]] == ["0.0", "0.06264832417874368", "0.1250505236945281",
                                                  "0.18696144082725336"]
    d["ka_sinewave440"] = ka[:, 0]
    save("osc.npz", **d)
    files["osc.npz"] = "maxiOsc, 12 waveforms, V=32, 2x160 samples; fm variants; KAT sinewave(440)"

    # ---- maxiFilter ---------------------------------------------------------------------------
    V, N = 32, 256
    rng = np.random.default_rng(SEED + 2)
    x = rng.uniform(-1, 1, (N, V))
    cutoff = np.concatenate([[5.0, 10.0, 50000.0, 22050.0], rng.uniform(20, 18000, V - 4)])
    res = np.concatenate([[0.5, 1.0, 30.0, 2.0], rng.uniform(1, 20, V - 4)])
    bres = rng.uniform(0.05, 1.2, V)
    lp = rng.uniform(0.0, 1.0, V)
    d = dict(x=x, cutoff=cutoff, res=res, bres=bres, lp=lp)
    for kind, name in enumerate(["lores", "hires", "bandpass", "lopass", "hipass"]):
        c = lp if kind >= 3 else cutoff
        r = None if kind >= 3 else (bres if kind == 2 else res)
        o1, st = R.filter(kind, x[:128], c, r)
        o2, st = R.filter(kind, x[128:], c, r, state=st)
        d["out_" + name] = np.concatenate([o1, o2])
        d["state_" + name] = st
        if kind <= 2:
            d["coef_" + name] = R.filter_coeffs(kind, c, r)
    cm = cutoff[None, :] * (1.0 + 0.5 * rng.uniform(-1, 1, (N, V)))
    d["cutoff_mod"] = cm
    o, st = R.filter(0, x, cm, res, cps=True)
    d["out_lores_mod"] = o
    save("filter.npz", **d)
    files["filter.npz"] = "maxiFilter 5 kinds, V=32, 2x128 samples + per-sample cutoff lores"

    # ---- maxiEnv ---------------------------------------------------------------------------------
    V, N = 16, 3000
    rng = np.random.default_rng(SEED + 3)
    att = np.array([R.env_coeff(0, ms) for ms in np.concatenate([[0.0, 1.0, 10.0], rng.uniform(0.5, 30, V - 3)])])
    dec = np.array([R.env_coeff(1, ms) for ms in rng.uniform(1, 40, V)])
    sus = rng.uniform(0.05, 0.9, V)
    rel = np.array([R.env_coeff(2, ms) for ms in rng.uniform(2, 30, V)])
    par = np.stack([att, dec, sus, rel])
    hold = np.concatenate([[1, 1, 0, 5], rng.integers(1, 200, V - 4)]).astype(np.int64)
    trig = ((np.arange(N) % 1500) < 700).astype(np.int32)
    trig_v = (rng.uniform(0, 1, (N, V)) < 0.002).astype(np.int32)
    xin = rng.uniform(-1, 1, (N, V))
    d = dict(par=par, hold=hold, trig=trig, trig_v=trig_v, xin=xin,
             setter_ms=np.array([0.0, 1.0, 10.0, 100.0, 500.0, 1000.0, 2000.0]))
    d["setters"] = np.array([[R.env_coeff(w, ms) for ms in d["setter_ms"]] for w in range(4)])
    for mode, name in enumerate(["adsr", "ar"]):
        o, dst, ist = R.env(mode, None, trig, par, hold)
        d["out_%s_gate" % name], d["dst_%s_gate" % name], d["ist_%s_gate" % name] = o, dst, ist
        o, dst, ist = R.env(mode, xin, trig_v, par, hold)
        d["out_%s_pv" % name], d["dst_%s_pv" % name], d["ist_%s_pv" % name] = o, dst, ist
    save("env.npz", **d)
    files["env.npz"] = "maxiEnv adsr/ar, V=16, 3000 samples, gate + per-voice impulse triggers; setters"

    # ---- fused subtractive voice (config 3, reduced) ----------------------------------------------
    V, N = 16, 4096
    v = np.arange(V) * 4096
    freq = np.minimum(20 + v * 0.30517578125, 5000.0)
    cutoff = 200 + 4 * freq
    res = 1.0 + (np.arange(V) % 16)
    par = np.stack([np.full(V, R.env_coeff(0, 10)), np.full(V, R.env_coeff(1, 100)), np.full(V, 0.5),
                    np.full(V, R.env_coeff(2, 500))])
    hold = np.ones(V, np.int64)
    trig = ((np.arange(N) % 2048) < 1024).astype(np.int32)
    d = dict(freq=freq, cutoff=cutoff, res=res, par=par, hold=hold, trig=trig)
    for mode in (0, 1):
        cu = cutoff if mode == 0 else np.full(V, 10000.0)
        o, ost, fst, dst, ist = R.voice(mode, freq, cu, res, trig, par, hold)
        d["out_mode%d" % mode] = o
        d["ost_mode%d" % mode], d["fst_mode%d" % mode] = ost, fst
        d["dst_mode%d" % mode], d["ist_mode%d" % mode] = dst, ist
    d["coef"] = R.filter_coeffs(0, cutoff, res)
    save("voice.npz", **d)
    files["voice.npz"] = "saw->lores->adsr voice, V=16, 4096 samples, mode A (hoisted) and B (modulated)"

    # ---- maxiMix::stereo mixdown ----------------------------------------------------------------
    V, N = 96, 64
    rng = np.random.default_rng(SEED + 4)
    x = rng.uniform(-1, 1, (N, V))
    pan = np.concatenate([[-0.5, 0.0, 1.0, 1.5], rng.uniform(0, 1, V - 4)])
    save("mix.npz", x=x, pan=pan, mix=R.mix_stereo(x, pan))
    files["mix.npz"] = "maxiMix::stereo + sequential voice sum, V=96, 64 samples"

    # ---- maxiDelayline ------------------------------------------------------------------------------
    V, N, cap = 24, 600, 96
    rng = np.random.default_rng(SEED + 5)
    x = rng.uniform(-1, 1, (N, V))
    size = np.concatenate([[1, 2, cap, 37], rng.integers(3, cap + 1, V - 4)]).astype(np.int32)
    size[12:] = 64  # half the bank shares one size (the coalesced case)
    fb = rng.uniform(0.0, 0.95, V)
    pos = rng.integers(0, cap, V).astype(np.int32)
    d = dict(x=x, size=size, fb=fb, pos=pos, cap=cap)
    for mode, name in enumerate(["dl", "dlFromPosition"]):
        o1, mem, ph = R.delay(mode, x[:250], size, fb, cap, position=pos)
        o2, mem, ph = R.delay(mode, x[250:], size, fb, cap, position=pos, mem=mem, phase=ph)
        d["out_" + name], d["mem_" + name], d["phase_" + name] = np.concatenate([o1, o2]), mem, ph
    save("delay.npz", **d)
    files["delay.npz"] = "maxiDelayline dl/dlFromPosition, V=24, 250+350 samples, cap 96"

    # ---- maxiSample play family ------------------------------------------------------------------------
    V, N, Ls = 24, 400, 1500
    rng = np.random.default_rng(SEED + 6)
    n = np.arange(Ls)
    smp = np.round((0.5 * np.sin(2 * np.pi * 110 * n / 44100) + 0.25 * np.sin(2 * np.pi * 331 * n / 44100)
                    + 0.05 * rng.uniform(-1, 1, Ls)) * 32767) / 32767.0  # int16-normalised like C:679
    d = dict(samples=smp, N=N)
    modes = ["play", "playOnce", "playLoop", "playUntil", "playAtSpeed", "playOnceAtSpeed",
             "playUntilAtSpeed", "play4", "playAtSpeedBetweenPoints"]
    for mode, name in enumerate(modes):
        pos0 = rng.uniform(1, Ls - 5, V)
        pos0[0] = Ls - 1.0  # the state setSample() leaves (H:677)
        if mode in (0, 1):
            pos0 = np.floor(pos0)
        a = rng.uniform(0.2, 3.0, V)
        st, en = rng.uniform(0, 0.4, V), rng.uniform(0.5, 1.1, V)
        if mode == 2:
            en = np.minimum(en, 1.0)  # playLoop does not clamp `end` (C:960-967): > 1 reads past the buffer
        if mode in (7, 8):
            a = rng.uniform(0.5, 60, V) * np.where(rng.uniform(0, 1, V) < 0.4, -1, 1)
            st, en = np.floor(rng.uniform(2, 400, V)), np.floor(rng.uniform(500, Ls - 1, V))
        o1, p = R.sample(mode, smp, N // 2, pos0, a=a, start=st, end=en)
        o2, p = R.sample(mode, smp, N - N // 2, p, a=a, start=st, end=en)
        d["pos0_" + name], d["a_" + name], d["start_" + name], d["end_" + name] = pos0, a, st, en
        d["out_" + name], d["pos_" + name] = np.concatenate([o1, o2]), p
    # per-sample speed modulation + a non-44100 mySampleRate (integer quotient C:1070) at sr 96000
    sp = rng.uniform(0.1, 2.5, (N, V))
    R.settings(96000, 2, 1024)
    o, p = R.sample(4, smp, N, np.zeros(V), a=sp, aps=True, mySampleRate=44100)
    R.settings(44100, 2, 1024)
    d["speed_mod"], d["out_speed_mod_sr96k"], d["pos_speed_mod_sr96k"] = sp, o, p
    save("sample.npz", **d)
    files["sample.npz"] = "maxiSample 9 play modes, V=24, 200+200 samples over a 1500-sample buffer"

    # ---- maxiFFT + maxiMFCC (config 4 signal, reduced) --------------------------------------------------
    rng = np.random.default_rng(SEED + 7)
    nsig = 1024 * 8
    n = np.arange(nsig)
    k = n // 1024
    sig = (0.4 * np.sin(2 * np.pi * 220 * n / 44100) + 0.3 * np.sin(2 * np.pi * (440 + 0.01 * k) * n / 44100)
           + 0.1 * rng.uniform(-1, 1, nsig)).astype(np.float32)
    d = dict(signal=sig)
    for (fs, hop, win) in [(1024, 1024, 1024), (1024, 256, 0), (512, 128, 512), (2048, 1024, 2048), (64, 64, 64)]:
        r = R.fft_stream(sig[:fs * 4 if fs > 1024 else nsig // (2 if hop < 1024 else 1)], fs, hop, win)
        tag = "%d_%d" % (fs, hop)
        for key in ("real", "imag", "mags", "phases"):
            d["%s_%s" % (key, tag)] = r[key]
    mags = d["mags_1024_1024"]
    d["db_1024_1024"] = R.fft_to_db(mags[0])
    for (nf, nc) in [(42, 13), (256, 13), (40, 20)]:
        mel, mf = R.mfcc(mags, nf, nc, 20.0, 20000.0)
        d["melbands_%d_%d" % (nf, nc)], d["mfcc_%d_%d" % (nf, nc)] = mel, mf
    save("spectral.npz", **d)
    files["spectral.npz"] = ("maxiFFT (1024/1024, 1024/256 streaming, 512/128, 2048/1024, 64/64) real/imag/mags/"
                             "phases + maxiMFCC 512/42/13, 512/256/13, 512/40/20 on the config-4 signal")

    # ---- maxiGrains: maxiTimeStretch / maxiStretch (config 5, reduced) ------------------------------------
    rng = np.random.default_rng(SEED + 8)
    Ls = 44100
    n = np.arange(Ls)
    smp = 0.5 * np.sin(2 * np.pi * 110 * n / 44100) + 0.25 * np.sin(2 * np.pi * 331 * n / 44100) \
        + 0.05 * rng.uniform(-1, 1, Ls)
    S, T = 12, 5000
    speed = 0.25 + 1.5 * (np.arange(S) % 97) / 96
    speed[5] = -0.8
    st0 = np.zeros((4, S))
    st0[0] = np.clip(np.arange(S) / S * Ls, 0, Ls - 1)  # setPosition(s/S)
    rnd = rng.integers(0, 10, (S, 48)).astype(np.int32)
    ts = rng.uniform(0.3, 1.7, S)
    d = dict(samples=smp, speed=speed, st0=st0, rnd=rnd, timestretch=ts, T=T)
    d["windows_2205"] = np.stack([R.grain_window(k, 2205) for k in range(9)])
    cases = {"ts_hann": (0, 0, 0.05, 4, True), "ts_hamming_norand": (0, 1, 0.03, 3, False),
             "st_hann": (1, 0, 0.05, 2, True), "st_gauss": (1, 8, 0.021, 5, True)}
    for name, (mode, w, gl, ov, use_rnd) in cases.items():
        h = T // 2
        o1, st, gst, rc = R.granular(mode, w, smp, h, speed, b=ts, rnd=rnd if use_rnd else None,
                                     grainLength=gl, overlaps=ov, st=st0)
        assert rc == 0
        o2, st, gst, rc = R.granular(mode, w, smp, T - h, speed, b=ts, rnd=rnd if use_rnd else None,
                                     grainLength=gl, overlaps=ov, st=st, gst=gst)
        assert rc == 0
        d["out_" + name], d["st_" + name], d["gst_" + name] = np.concatenate([o1, o2]), st, gst
    save("grains.npz", **d)
    files["grains.npz"] = "maxiTimeStretch/maxiStretch banks, 12 streams x 2500+2500 samples, 4 configs; 9 windows"

    # ---- rows closed later in round 1: noise, quad/ambisonic, trigger-driven maxiSample players,
    # ---- playWithPhasor, FFT features, playAtPosition, maxiPitchShift ------------------------------------
    rng = np.random.default_rng(SEED + 9)
    d = {}
    V, N = 24, 96
    d["noise_seed"] = 20260923
    d["noise_rand"], d["noise_out"] = R.noise(int(d["noise_seed"]), V, N)
    x = rng.uniform(-1, 1, (N, V))
    px, py, pz = rng.uniform(-0.2, 1.2, V), rng.uniform(-0.2, 1.2, V), rng.uniform(-0.3, 1.3, V)
    d["bus_x"], d["bus_px"], d["bus_py"], d["bus_pz"] = x, px, py, pz
    for C in (2, 4, 8):
        d["mix_%d" % C], d["bus_%d" % C] = R.mix_bus(C, x, px, py, pz, want_bus=True)
    Ls = 1500
    n = np.arange(Ls)
    smp = np.round((0.5 * np.sin(2 * np.pi * 110 * n / 44100) + 0.25 * np.sin(2 * np.pi * 331 * n / 44100)
                    + 0.05 * rng.uniform(-1, 1, Ls)) * 32767) / 32767.0
    N = 300
    trig = np.sin(np.arange(N)[:, None] * rng.uniform(0.02, 0.3, V)[None, :] + rng.uniform(0, 6, V))
    trig[:, ::5] = np.where(rng.uniform(size=(N, (V + 4) // 5)) < 0.05, 1.0, 0.0)
    d["smp"], d["zx_trig"] = smp, trig
    d["zx_a"], d["zx_p0"], d["zx_p1"] = rng.uniform(0.3, 2.5, V), rng.uniform(0, 0.6, V), rng.uniform(0.1, 0.5, V)
    d["zx_pos0"] = rng.uniform(0, Ls - 1, V)
    for mode in range(5):
        h = N // 2
        o1, p, zp, zf = R.sample_zx(mode, smp, trig[:h], d["zx_pos0"], a=d["zx_a"], p0=d["zx_p0"], p1=d["zx_p1"])
        o2, p, zp, zf = R.sample_zx(mode, smp, trig[h:], p, a=d["zx_a"], p0=d["zx_p0"], p1=d["zx_p1"],
                                    zx_prev=zp, zx_first=zf)
        d["zx_out_%d" % mode], d["zx_pos_%d" % mode] = np.concatenate([o1, o2]), p
        d["zx_prev_%d" % mode], d["zx_first_%d" % mode] = zp, zf
    pha = (np.arange(N)[:, None] * rng.uniform(0.0003, 0.01, V)[None, :] + rng.uniform(0, 1, V)) % 1.0
    pha[:, 1::4] = 1.0 - pha[:, 1::4]
    pha[:, 2::4] = np.round(pha[:, 2::4] * 8) / 8
    pha[150:153] = rng.uniform(-0.3, 1.3, (3, V))
    pha[:, 3] = 0.0
    pha[1:, 7] = np.linspace(0.001, 0.0, N - 1)
    d["phasor_in"] = pha
    o1, pp, pf = R.sample_phasor(smp, pha[:150])
    o2, pp, pf = R.sample_phasor(smp, pha[150:], phasor_prev=pp, phasor_first=pf)
    d["phasor_out"], d["phasor_prev"], d["phasor_first"] = np.concatenate([o1, o2]), pp, pf
    m = np.abs(rng.normal(0, 30, (40, 512))).astype(np.float32)
    m[5] = 0
    m[6, ::3] = 0
    m[7, :256] = 5e-7
    d["feat_mags"] = m
    d["feat_flatness"], d["feat_centroid"] = R.fft_features(m, 1024)
    d["feat_db"] = R.fft_to_db(m)
    Ls = 30000
    gsmp = rng.uniform(-1, 1, Ls)
    S, T = 12, 3000
    pos = ((np.arange(T)[:, None] * rng.uniform(0.2, 2.0, S)[None, :] / Ls) + rng.uniform(0, 1, S)) % 1.0
    pos[:, 3] = rng.uniform(-0.2, 1.2, T)
    pos[:, 5] = 1.0
    d["g_samples"], d["pap_pos"] = gsmp, pos
    h = T // 2
    o1, st, gst, rc = R.granular(2, 0, gsmp, h, pos[:h], grainLength=0.05, overlaps=4)
    assert rc == 0
    o2, st, gst, rc = R.granular(2, 0, gsmp, T - h, pos[h:], grainLength=0.05, overlaps=4, st=st, gst=gst)
    assert rc == 0
    d["pap_out"], d["pap_st"], d["pap_gst"] = np.concatenate([o1, o2]), st, gst
    speed = rng.uniform(-2.0, 2.5, S)
    speed[:4] = [1.0, 0.5, -1.0, 2.0]
    pm = rng.uniform(-0.1, 0.1, S)
    st0 = np.zeros((4, S))
    st0[0] = rng.uniform(0, Ls, S)
    st0[0, 7] = Ls - 3.0
    d["ps_speed"], d["ps_posmod"], d["ps_st0"] = speed, pm, st0
    o1, st, gst, rc = R.granular(3, 0, gsmp, h, speed, posMod=pm, grainLength=0.05, overlaps=3, st=st0)
    assert rc == 0
    o2, st, gst, rc = R.granular(3, 0, gsmp, T - h, speed, posMod=pm, grainLength=0.05, overlaps=3, st=st, gst=gst)
    assert rc == 0
    d["ps_out"], d["ps_st"], d["ps_gst"] = np.concatenate([o1, o2]), st, gst
    save("extra.npz", **d)
    files["extra.npz"] = ("maxiOsc::noise (srand seed + draws), maxiMix stereo/quad/ambisonic bus, playOnZX* x5, "
                          "playWithPhasor, magsToDB/spectralFlatness/spectralCentroid, playAtPosition, maxiPitchShift")

    # ---- maxiSample::load/save: 16-bit PCM WAV fixtures (synthesised here, never the reference's assets) ----
    import struct
    wavdir = os.path.join(GOLD, "wav")
    os.makedirs(wavdir, exist_ok=True)
    rng = np.random.default_rng(SEED + 10)

    def wav_bytes(data_i16, channels=1, rate=44100, extra=b"", fmt_extra=b""):
        data = data_i16.astype("<i2").tobytes()
        fmt = struct.pack("<HHIIHH", 1, channels, rate, rate * channels * 2, channels * 2, 16) + fmt_extra
        body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + extra + b"data" + struct.pack("<I", len(data)) + data
        return b"RIFF" + struct.pack("<I", len(body)) + body

    x = (rng.uniform(-1, 1, 3001) * 32767).astype(np.int16)
    x[:4] = [32767, -32768, 0, 1]
    st = (rng.uniform(-1, 1, 2000) * 32767).astype(np.int16)
    wavs = {"mono": wav_bytes(x), "list": wav_bytes(x, extra=b"LIST" + struct.pack("<I", 10) + b"INFOabcdef"),
            "fmt18": wav_bytes(x, fmt_extra=b"\x00\x00", rate=22050), "stereo": wav_bytes(st, channels=2)}
    d = {}
    if not ONLY or "wav.npz" in ONLY:
        for k, b in wavs.items():
            with open(os.path.join(wavdir, k + ".wav"), "wb") as f:
                f.write(b)
    for k in wavs:
        for ch in ((0, 1) if k == "stereo" else (0,)):
            amp, hdr, pos = R.wav_load(os.path.join(wavdir, k + ".wav"), ch)
            n = amp.size
            if k == "stereo":   # beyond the defined prefix the reference reads past its vector (C:669-672)
                d["defined_%s_%d" % (k, ch)] = len(range(ch * 2, n, 4))
            d["amp_%s_%d" % (k, ch)], d["hdr_%s_%d" % (k, ch)], d["pos_%s_%d" % (k, ch)] = amp, hdr, pos
    amp = rng.uniform(-1, 1, 777)
    amp[:7] = [1.0, -1.0, 0.5 / 32767, 1.5 / 32767, -0.5 / 32767, 2.5 / 32767, -32768 / 32767.0]
    hdr = np.array([36 + 777 * 2, 16, 1, 1, 44100, 88200, 2, 16], np.int32)
    d["save_amp"], d["save_hdr"] = amp, hdr
    if not ONLY or "wav.npz" in ONLY:
        R.wav_save(os.path.join(wavdir, "saved_by_reference.wav"), amp, hdr)
    save("wav.npz", **d)
    files["wav.npz"] = ("maxiSample::load of wav/{mono,list,fmt18,stereo}.wav (amplitudes, header fields, position) and the "
                        "input of wav/saved_by_reference.wav (maxiSample::save)")

    # ---- maxiIFFT (SPECTRUM mode), two consecutive calls with the overlap-add buffer carried -----------------
    rng = np.random.default_rng(SEED + 11)
    d = {}
    for (fs, hop, win) in [(1024, 512, 0), (1024, 256, 1024), (64, 16, 48)]:
        nf = 7
        m = np.abs(rng.normal(0, 3, (nf, fs // 2))).astype(np.float32)
        ph = rng.uniform(-np.pi, np.pi, (nf, fs // 2)).astype(np.float32)
        ph[:, ::3] = 0.0
        tag = "%d_%d_%d" % (fs, hop, win)
        o1, io1, buf = R.ifft_stream(m[:3], ph[:3], fs, hop, win)
        o2, io2, buf = R.ifft_stream(m[3:], ph[3:], fs, hop, win, buffer=buf)
        d["mags_" + tag], d["phases_" + tag] = m, ph
        d["signal_" + tag], d["ifftout_" + tag], d["buffer_" + tag] = np.concatenate([o1, o2]), np.concatenate([io1, io2]), buf
    save("ifft.npz", **d)
    files["ifft.npz"] = "maxiIFFT 1024/512, 1024/256/1024, 64/16/48: signal, per-frame ifftOut, final overlap-add buffer"

    # ---- maxiDCBlocker / maxiSVF / maxiBiquad / maxiEnvGen / maxiSampler -----------------------------------
    rng = np.random.default_rng(SEED + 12)
    d = {}
    V, N = 21, 240
    x = rng.uniform(-1, 1, (N, V))
    d["f2_x"] = x
    par = rng.uniform(0.9, 0.9999, (1, V))
    o1, st, _ = R.filter2(0, x[:100], par)
    o2, st, _ = R.filter2(0, x[100:], par, st)
    d["dc_par"], d["dc_out"], d["dc_st"] = par, np.concatenate([o1, o2]), st
    par = np.stack([rng.uniform(20, 20000, V), rng.uniform(0, 12, V), *rng.uniform(0, 1, (4, V))])
    par[1, :2] = 0
    o1, st, c = R.filter2(1, x[:100], par)
    o2, st, c = R.filter2(1, x[100:], par, st)
    d["svf_par"], d["svf_out"], d["svf_st"], d["svf_coef"] = par, np.concatenate([o1, o2]), st, c
    for t in range(7):
        par = np.stack([np.full(V, float(t)), rng.uniform(30, 18000, V), rng.uniform(0.3, 8, V), rng.uniform(-18, 18, V)])
        o1, st, c = R.filter2(2, x[:100], par)
        o2, st, c = R.filter2(2, x[100:], par, st)
        d["bq_par_%d" % t], d["bq_out_%d" % t], d["bq_st_%d" % t], d["bq_coef_%d" % t] = par, np.concatenate([o1, o2]), st, c
    H = R.ENVGEN_HOLD
    N = 3000
    n = np.arange(N)[:, None]
    trig = np.sign(np.sin(n * rng.uniform(0.002, 0.01, V)[None, :] + rng.uniform(0, 6, V)))
    trig[:, 3] = 1.0
    trig[:, 4] = (np.arange(N) % 700 < 5) * 1.0
    d["eg_trig"] = trig
    shapes = {"ar": ([0, 1, 0], [10, 40], [1, 1]), "adsr": ([0, 1, 0.4, 0.4, 0], [3, 12, H, 25], [1, 1, 1, 1]),
              "curved": ([0, 1, 0.2, 0], [7.3, 11.1, 20.7], [0.5, 2, 3])}
    for name, (lv, tm, cv) in shapes.items():
        for loop, retrig in ((0, 0), (1, 1)):
            o1, ds, is_, stg = R.envgen(trig[:1400], lv, tm, cv, loop, retrig)
            o2, ds, is_, _ = R.envgen(trig[1400:], lv, tm, cv, loop, retrig, dst=ds, ist=is_)
            tag = "%s_%d%d" % (name, loop, retrig)
            d["eg_out_" + tag], d["eg_dst_" + tag], d["eg_ist_" + tag] = np.concatenate([o1, o2]), ds, is_
        d["eg_levels_" + name], d["eg_times_" + name], d["eg_curves_" + name] = np.array(lv, float), np.array(tm, float), np.array(cv, float)
        d["eg_stages_" + name] = stg
    voices, NS, N = 8, 3, 900
    Vs = voices * NS
    ssmp = rng.uniform(-1, 1, 3000)
    d["smp_samples"], d["smp_pitch"], d["smp_gain"] = ssmp, rng.integers(-24, 25, Vs).astype(np.float64), rng.uniform(0.2, 1.0, Vs)
    d["smp_par"] = np.stack([rng.uniform(0.001, 0.2, Vs), rng.uniform(0.99, 0.9999, Vs), rng.uniform(0.3, 1.0, Vs), rng.uniform(0.99, 0.9999, Vs)])
    d["smp_hold"] = rng.integers(1, 50, Vs)
    d["smp_trig0"] = (rng.uniform(size=Vs) < 0.6).astype(np.int32)
    for sustain in (1, 0):
        a = R.sampler(voices, ssmp, N, d["smp_pitch"], d["smp_gain"], d["smp_par"], d["smp_hold"], np.zeros(Vs), d["smp_trig0"], sustain)
        t2 = a[3].copy()
        t2[::3] = 0
        t2[1::7] = 1
        b = R.sampler(voices, ssmp, N, d["smp_pitch"], d["smp_gain"], d["smp_par"], d["smp_hold"], a[2], t2, sustain, a[4], a[5], a[6])
        for k, nm in enumerate(["mix", "outputs", "position", "trigger", "outhold", "dst", "ist"]):
            d["smp_%s_a%d" % (nm, sustain)], d["smp_%s_b%d" % (nm, sustain)] = a[k], b[k]
    save("extra2.npz", **d)
    files["extra2.npz"] = ("maxiDCBlocker, maxiSVF, maxiBiquad x7 types (outputs, state, coefficients), maxiEnvGen AR/ADSR/curved x "
                           "loop/retrigger (outputs, detector + stage state, stage tables), maxiSampler 3x8 slots with note-offs")

    # ---- maxiConvolve: impulse analysis + play(), as the reference computes (mode 0) and as intended (mode 1) ----
    rng = np.random.default_rng(SEED + 77)
    d = {}
    for tag, (Li, F, H) in {"a": (5000, 1024, 256), "b": (1800, 256, 64), "c": (2049, 1024, 256)}.items():
        pcm = (rng.uniform(-1, 1, Li) * np.exp(-np.arange(Li) / (Li / 6.0)) * 24000).astype(np.int16)
        x = rng.uniform(-1, 1, F * 9).astype(np.float32)
        d["pcm_" + tag], d["x_" + tag], d["cfg_" + tag] = pcm, x, np.array([F, H])
        for mode in (0, 1):
            o, ir, ii = R.convolve(pcm, x, F, H, mode)
            d["out%d_%s" % (mode, tag)] = o
        d["impR_" + tag], d["impI_" + tag] = ir, ii
    save("convolve.npz", **d)
    files["convolve.npz"] = ("maxiConvolve setup (impulse frames, normalised) + play() over 9 input frames for 1024/256 (5 and 2 "
                             "impulse frames) and 256/64: mode 0 = the reference verbatim (silence), mode 1 = sums routed to the "
                             "inverse transform's inputs through the reference's own calcIFFT")

    # ---- the reference's own example patches through the headless host (oracle/example_host.cpp): the oracle of the
    # drop-in header include/maximilian.h.  01 = cpp/commandline/main.cpp (BASELINE config 1: 1x sinewave(440), 44 100
    # frames x 2 ch), 14 = 14.monosynth (96 000 frames: the first metronome tick falls at 88 200), 15 = 15.polysynth.
    d = {}
    for ex, frames in (("01", 44100), ("14", 96000), ("15", 16384)):
        tmp = os.path.join("/tmp", "mxo_example_%s.f64" % ex)
        subprocess.run([os.path.join(HERE, "_ref", "example_" + ex), str(frames), tmp], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        d["ex" + ex] = np.fromfile(tmp, np.float64).reshape(frames, 2)
        os.remove(tmp)
    save("dropin.npz", **d)
    # ten more of the reference's example patches (maximilian_examples/2 .. 13), 6000 frames each, left channel only (they write
    # the same value to both, 13.Advanced-Filters writes channel 0 only)
    d2 = {}
    for ex in ("02", "03", "04", "05", "06", "08b", "08c", "08d", "10", "11", "13", "16"):
        tmp = os.path.join("/tmp", "mxo_example_%s.f64" % ex)
        frames = 30000 if ex == "16" else 6000   # 16.Replicant: six of its 9 Hz metronome ticks
        subprocess.run([os.path.join(HERE, "_ref", "example_" + ex), str(frames), tmp], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        d2["ex" + ex] = np.fromfile(tmp, np.float64).reshape(frames, 2)[:, 0].copy()
        os.remove(tmp)
    # 12.SamplePlayer loads "../../../beat2.wav" relative to the working directory: the golden run (and the GPU test) put
    # tests/golden/wav/mono.wav (3001 samples: the head wraps inside the 6000 frames) there under that name
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        cwd = os.path.join(td, "a", "b", "c")
        os.makedirs(cwd)
        shutil.copy(os.path.join(GOLD, "wav", "mono.wav"), os.path.join(td, "beat2.wav"))
        tmp = os.path.join(td, "ex12.f64")
        subprocess.run([os.path.join(HERE, "_ref", "example_12"), "6000", tmp], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, cwd=cwd)
        d2["ex12"] = np.fromfile(tmp, np.float64).reshape(6000, 2)[:, 0].copy()
        # 20.FFT_example over the same file: play() -> maxiFFT(1024, 512, 1024) -> bins shifted by a looping maxiEnvGen -> maxiIFFT
        tmp = os.path.join(td, "ex20.f64")
        subprocess.run([os.path.join(HERE, "_ref", "example_20"), "8192", tmp], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, cwd=cwd)
        d2["ex20"] = np.fromfile(tmp, np.float64).reshape(8192, 2)[:, 0].copy()
    # tests/patches/filters2_patch.cpp (maxiSVF, maxiBiquad x3, maxiDCBlocker, maxiEnvGen ADSR gated by an oscillator) built against
    # the reference: both channels (output, envelope), 20 000 frames = trigger, attack, decay, hold and release
    tmp = os.path.join("/tmp", "mxo_example_p1.f64")
    subprocess.run([os.path.join(HERE, "_ref", "example_p1"), "20000", tmp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p1 = np.fromfile(tmp, np.float64).reshape(20000, 2)
    os.remove(tmp)
    d2["exp1"], d2["exp1_env"] = p1[:, 0].copy(), p1[:, 1].copy()
    # the reference's own test programs of this path (cpp/commandline/tests/{ffttest,mfcctest,svftest}), 8192 frames of channel 0;
    # mfcctest prints mfccs[1] once per frame: the printed numbers are kept too
    for tag in ("tfft", "tmfcc", "tsvf"):
        tmp = os.path.join("/tmp", "mxo_example_%s.f64" % tag)
        r = subprocess.run([os.path.join(HERE, "_ref", "example_" + tag), "8192", tmp], check=True, stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, text=True)
        d2["ex" + tag] = np.fromfile(tmp, np.float64).reshape(8192, 2)[:, 0].copy()
        os.remove(tmp)
        if tag == "tmfcc":
            d2["extmfcc_printed"] = np.array([float(x) for x in r.stdout.replace("Setup", "").split()])
    save("dropin_examples.npz", **d2)
    files["dropin_examples.npz"] = ("maximilian_examples 2.TwoTones, 3.AM1, 4.AM2, 5.FM1, 6.FM2, 8.Counting2/3/4, 10.Filters, 11.Mixing, 12.SamplePlayer and 20.FFT_example (8192 frames; both over tests/golden/wav/mono.wav), 13.Advanced-Filters, 16.Replicant of "
                                    "the reference: 6000 frames (16: 30000) of channel 0 each, compiled with the unmodified reference library; exp1 / exp1_env: tests/patches/filters2_patch.cpp the same way (20000 frames, both channels); extfft / extmfcc (+ the mfccs[1] it prints per frame) / extsvf: the reference's own cpp/commandline/tests ffttest, mfcctest, svftest, 8192 frames")
    files["dropin.npz"] = ("cpp/commandline/main.cpp (44100 frames), 14.monosynth (96000), 15.polysynth (16384) of the reference, "
                           "compiled with the unmodified reference library and run through oracle/example_host.cpp (routing() restated)")

    sha = hashlib.sha256()
    for f in ("libs/maxiConvolve.cpp", "libs/maxiConvolve.h", "libs/maxiFFT.h", "libs/fft.h", "maximilian.cpp", "maximilian.h", "libs/fft.cpp", "libs/maxiFFT.cpp", "libs/maxiMFCC.cpp",
              "libs/maxiMFCC.h", "libs/maxiGrains.h", "libs/maxiSynths.cpp", "libs/maxiSynths.h"):
        sha.update(open(os.path.join(REF_SRC, f), "rb").read())
    manifest = {
        "generator": "oracle/gen_golden.py",
        "source": "oracle/_ref/libmaxiref.so (unmodified /root/reference sources + oracle/ref_harness.cpp)",
        "reference_sources_sha256": sha.hexdigest(),
        "compiler": subprocess.run(["g++", "--version"], capture_output=True, text=True).stdout.splitlines()[0],
        "flags": "-std=c++17 -O2 -ffp-contract=off -fno-fast-math (no -march=native)",
        "libc": " ".join(platform.libc_ver()),
        "seed": hex(SEED),
        "files": files,
    }
    with open(os.path.join(GOLD, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    for k in sorted(os.listdir(GOLD)):
        print(k, os.path.getsize(os.path.join(GOLD, k)))


if __name__ == "__main__":
    main()
