"""DROP-IN MODE (include/skred_synth_abi.h, libskred_synth.so).

CPU: the library exports the synth.h surface; the reference's own wire()/seq()/skred.c objects link
against it (oracle/Makefile: dropin_check) and every control line of every golden case leaves our
arrays bit-identical to what the reference's synth.o produced.
GPU (-m gpu): the same replay with the audio callback running: reference synth_callback() ->
our synth() -> HIP kernels, compared with the reference's output (stems/state bit-exact, mix 1e-5).
"""
import json
import os
import re
import subprocess
import sys

import pytest

import golden_io as gio

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SYNTH_SO = os.path.join(ROOT, "skred_amd", "libskred_synth.so")
CHECK_SO = os.path.join(ROOT, "oracle", "_ref", "libskred_dropin_check.so")
REFERENCE = os.environ.get("SKRED_REFERENCE", "/root/reference")


def declared():
    text = open(os.path.join(ROOT, "include", "skred_synth_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    data = re.findall(r"^extern\s+[^;]*?\**\s*(\w+)\s*(?:\[[^\]]*\])?\s*;", text, flags=re.M)
    multi = re.findall(r"^extern float (volume_user, [^;]+);", text, flags=re.M)
    for m in multi:
        data += [x.strip() for x in m.split(",")]
    data += ["volume_threshold", "volume_smoother_higher_smoothing"]
    funcs = re.findall(r"^\s*(?:[\w\*]+\s+)+\**(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M)
    return sorted(set(data)), sorted(set(funcs))


def reference_surface():
    """Every function synth.h prototypes, every scalar it declares extern and every synth.def array, read from the
    reference tree itself (SURVEY 8b "What a C-ABI replacement must export")."""
    hdr = re.sub(r"/\*.*?\*/|//[^\n]*", "", open(os.path.join(REFERENCE, "synth.h")).read(), flags=re.S)
    funcs = re.findall(r"^\s*(?:[\w\*]+\s+)+\**(\w+)\s*\([^;{]*\)\s*;", hdr, flags=re.M)
    scalars = re.findall(r"^extern\s+[^;(]*?(\w+)\s*;", hdr, flags=re.M)
    arrays = re.findall(r"^ARRAY\(\s*[^,]+,\s*(\w+)\s*,", open(os.path.join(REFERENCE, "synth.def")).read(), flags=re.M)
    return sorted(set(funcs)), sorted(set(scalars)), sorted(set(arrays))


def test_dropin_exports_synth_h_surface():
    import ctypes
    L = ctypes.CDLL(SYNTH_SO)
    data, funcs = declared()
    assert len(data) >= 75 + 9 and len(funcs) >= 53, (len(data), len(funcs))
    missing = [s for s in data + funcs if not hasattr(L, s)]
    assert not missing, missing
    if os.path.isdir(REFERENCE):
        rf, rs, ra = reference_surface()
        assert len(rf) >= 53 and len(rs) == 9 and len(ra) == 75, (len(rf), len(rs), len(ra))
        missing = [s for s in rf + rs + ra if not hasattr(L, s)]
        assert not missing, f"synth.h / synth.def symbols the drop-in does not export: {missing}"
        undeclared = [s for s in rf + rs + ra if s not in data + funcs]
        assert not undeclared, f"exported but not declared in include/skred_synth_abi.h: {undeclared}"


def persample(mode, case):
    out = subprocess.run([sys.executable, os.path.join(HERE, "persample_replay.py"), mode, case],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


@pytest.mark.parametrize("case", ["edge_mod", "edge_basic", "c4_pcm_oneshot", "c1_sine_adsr64", "korg_waves"])
def test_per_sample_functions_match_reference(case):
    """The eight per-sample functions of synth.h (osc_next, cz_phasor, quantize_bits_int, mmf_process,
    amp_envelope_step, audio_rng_*) and voice_format() of the drop-in against the compiled reference: same calls on the
    same voice state (the libraries' own tables incl. the Korg slots, then the fixture's), every returned float, the
    state left behind and the text, bit for bit."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libskred_ref.so")):
        pytest.skip("oracle/_ref/libskred_ref.so not built (needs the reference tree)")
    ref, mine = persample("ref", case), persample("mine", case)
    assert mine["stateless"] == ref["stateless"]
    assert mine["state_and_values"] == ref["state_and_values"]
    assert mine["text"] == ref["text"], (mine["text_sample"], ref["text_sample"])


def ensure_check_lib():
    if os.path.isdir(REFERENCE):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref", "dropin_check"], check=True)
    if not os.path.exists(CHECK_SO):
        pytest.skip("oracle/_ref/libskred_dropin_check.so not built (needs the reference tree)")


def replay(case, render):
    out = subprocess.run([sys.executable, os.path.join(HERE, "dropin_replay.py"), case] + (["--render"] if render else []),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.parametrize("case", gio.CASES)
def test_control_path_state_matches_reference(case):
    """reference wire() -> OUR setters == reference wire() -> reference setters, bit for bit."""
    ensure_check_lib()
    r = replay(case, render=False)
    s = r["segments"][0]
    assert s["state_in"] == {}, s
    assert s["tables_equal"] and s["globals_in_equal"], s


@pytest.mark.gpu
@pytest.mark.parametrize("case", gio.CASES)
def test_dropin_render_matches_reference(case):
    """reference synth_callback() -> OUR synth() -> GPU, against the reference's own output."""
    ensure_check_lib()
    r = replay(case, render=True)
    for s in r["segments"]:
        assert s.get("rc", 0) == 0, s
        assert s["state_in"] == {} and s["tables_equal"], s
        assert s["stems_sha_equal"], s
        assert s["state_out"] == {}, s
        assert s["mix_rms_err"] <= 1e-5, s
        assert s["count_out_equal"], s
    # e.g. wav_samples: the reference's recorder (skred.c:120-131, wire.c:94-185) fed by OUR stems
    assert all(r["extras"].values()), r["extras"]


@pytest.mark.gpu
def test_sk_render_config0_wav_matches_reference(tmp_path):
    """BASELINE config 0 end to end on the GPU: the C harness loads 0.sk with our patch reader, calls
    synth() in 512-frame callbacks and writes a float WAV; its samples equal the reference's 1 s render
    (one audible voice, so even the mix is bit-exact)."""
    import numpy as np
    exe = os.path.join(ROOT, "skred_amd", "sk_render")
    wav = str(tmp_path / "c0.wav")
    out = subprocess.run([exe, "--patch-0sk", "--seconds", "1", wav], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    raw = open(wav, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE"
    frames = np.frombuffer(raw[44:], dtype="<f4").reshape(-1, 2)
    gold = gio.load("c0_0sk").segments[0].mix
    assert frames.shape == gold.shape
    assert gio.bits_equal(frames, gold)
