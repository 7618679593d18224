"""SURVEY 8f "next" #3: WAV files in, stem recording out.

Truth is the compiled reference: tests/golden/wav_samples.npz holds the WAV bytes that were fed to the
reference's `:w` (wire.c:406-441 -> miniwav.c:103-147 -> miniaudio), the tables and slot fields it
installed, the audio it rendered from them, and the 6-channel 16-bit file its recorder wrote
(skred.c:120-131 -> wire.c:94-185 save_wav)."""
import os

import numpy as np
import pytest

import golden_io as gio
from skred_amd import wav


@pytest.fixture(scope="module")
def gold():
    return gio.load("wav_samples")


def test_header_symbols_exported():
    import ctypes
    L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                 "skred_amd", "libskred_amd.so"))
    missing = [s for s in wav.WAV_SYMBOLS if not hasattr(L, s)]
    assert not missing, missing
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "skred_wav.h")).read()
    assert all(s + "(" in hdr for s in wav.WAV_SYMBOLS)


def test_decoded_tables_equal_reference_decoder(gold):
    """u8 / s16 / s24 / s32 / f32, mono and stereo, extensible header, extra chunks, every channel choice."""
    x = gold.extras
    for k, (which, slot, ch) in enumerate(x["wav_loads"]):
        got = wav.wav_get(x[f"wav_in_{which}"].tobytes(), int(ch))
        assert got is not None, (which, slot, ch)
        table, info = got
        assert gio.bits_equal(table, x[f"slot_table_{slot}"]), (which, slot, ch)
        assert info.frames == x["slot_wave_size"][k]
        assert np.float32(info.sample_rate) == x["slot_wave_rate"][k]
        # wave_offset_hz as wave_load derives it (wire.c:436)
        assert np.float32(np.float32(info.frames) / np.float32(info.sample_rate) * np.float32(440.0)) == x["slot_wave_offset_hz"][k]


def test_wav_get_from_file_and_errors(gold, tmp_path):
    x = gold.extras
    p = tmp_path / "1.wav"
    p.write_bytes(x["wav_in_1"].tobytes())
    table, info = wav.wav_get(str(p), 0)
    assert info.channels == 1 and info.bits_per_sample == 16 and len(table) == 3000
    assert wav.wav_get(str(tmp_path / "missing.wav")) is None            # miniwav.c:113-117
    assert wav.wav_get(b"RIFF\x04\x00\x00\x00WAVE") is None               # no fmt / data
    assert wav.wav_get(b"not a wav file at all") is None
    raw = x["wav_in_1"].tobytes()
    cut, _ = wav.wav_get(raw[:44 + 2 * 100], 0)                          # truncated data chunk: what is there
    assert len(cut) == 100 and gio.bits_equal(cut, table[:100])


def test_default_channel_is_next_frames_channel0(gold):
    """The reference's default (`:wN,slot`, ch = -1) plays channel 0 advanced by one sample (miniwav.c:130,137:
    the -1 is compared as unsigned); its last element is an over-read, defined as 0.0f here."""
    raw = gold.extras["wav_in_2"].tobytes()
    ch0, _ = wav.wav_get(raw, 0)
    dflt, _ = wav.wav_get(raw, -1)
    assert gio.bits_equal(dflt[:-1], ch0[1:]) and dflt[-1] == 0.0
    assert gio.bits_equal(wav.wav_get(raw, 7)[0], dflt)                  # ch > channels: same selection


@pytest.mark.gpu
def test_recorder_file_equals_reference_save_wav(gold, tmp_path):
    """Bank-mode render of the case with stems -> device recorder -> save: byte-identical to the file the
    reference's `<0.05` ... `*` produced (header, min/max scale, truncating int16 conversion, voice selection)."""
    import torch
    from skred_amd import device
    seg = gold.segments[0]
    n = seg.bank_in.n
    db = device.DeviceBank(n, 0)
    db.set_tables(gold.tables)
    db.upload(seg.bank_in)
    db.set_globals(seg.g_in)
    rec = wav.Recorder(n, capacity_frames=44100)                        # synth_callback_init(1.0)
    rec.start(int(0.05 * 44100 * 2 * 64) // 128)                        # `<0.05`: rec_max floats -> whole frames
    stream = torch.cuda.current_stream().cuda_stream
    partial = torch.zeros(seg.block, 2, device="cuda")
    stems = torch.zeros(seg.block, n, 2, device="cuda")
    done = 0
    while done < seg.frames:
        f = min(seg.block, seg.frames - done)
        db.render(f, partial.data_ptr(), stems.data_ptr(), 0, stream)
        rec.append(stems.data_ptr(), f, stream)
        done += f
    torch.cuda.synchronize()
    assert not rec.recording and rec.frames == 2205
    record = np.zeros(n, np.int32)
    record[[0, 3, 5]] = 1                                               # `v0 r1`, `v3 r1`, `v5 r1`
    out = tmp_path / "take.wav"
    rec.save_wav(str(out), record)
    want = gold.extras["rec_wav"].tobytes()
    got = out.read_bytes()
    assert len(got) == len(want) == 44 + 2205 * 6 * 2
    assert got[:44] == want[:44]
    assert got == want
    # same samples through the buffer interface; an empty selection writes nothing
    pcm = rec.convert(record)
    assert pcm.tobytes() == want[44:]
    none = tmp_path / "none.wav"
    rec.save_wav(str(none), np.zeros(n, np.int32))
    assert not none.exists()
    rec.close()
    db.close()


@pytest.mark.gpu
def test_recorder_scan_and_convert_at_size():
    """2^24 samples of seeded noise with planted extremes: the device min/max scan and conversion equal the
    two loops of save_wav restated in numpy (exact: max/min are order-free, the conversion is per sample)."""
    import torch
    n_voices, frames = 256, 32768
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.rand(frames, n_voices, 2, device="cuda", generator=g) - 0.5) * 1.3
    x[12345, 17, 1] = -2.75
    x[999, 200, 0] = 1.9
    x[5, 5, 0] = float("nan")                                           # never wins a comparison
    rec = wav.Recorder(n_voices, frames)
    rec.start()
    rec.append(x.data_ptr(), frames)
    record = np.zeros(n_voices, np.int32)
    record[[3, 17, 200, 255]] = 1
    pcm = rec.convert(record).reshape(frames, 4, 2)
    h = x.cpu().numpy()
    scale = np.float32(-1.0) / np.float32(-2.75)                        # |min| > |max|
    sel = h[:, [3, 17, 200, 255], :] * scale
    sel = np.minimum(np.maximum(sel, np.float32(-1.0)), np.float32(1.0))
    want = np.nan_to_num(sel * np.float32(32767.0), nan=0.0).astype(np.int32).astype(np.int16)
    assert (pcm == want).all()
    rec.close()


def test_patch_reader_colon_w_matches_reference_state(gold, tmp_path):
    """`:wN,slot,ch` + the voice lines of the case through THIS build's patch reader (skred_patch.c ->
    skred_wave_load -> skred_wav_get): every per-voice field and every voice's table equal what the
    reference's wire() / wave_load / mw_get left behind.  Fresh process: the library holds global state."""
    import json
    import subprocess
    import sys
    x = gold.extras
    for which in range(1, 8):
        (tmp_path / f"{which}.wav").write_bytes(x[f"wav_in_{which}"].tobytes())
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "wav_patch_replay.py")], cwd=str(tmp_path),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1500:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert r["unsupported"] == 0 and r["errors"] == 0, r
    assert r["field_mismatches"] == {}, r
    assert r["tables_equal"] and r["slots_equal"], r
    assert r["bad_slot_rc"] == 17 and r["missing_file_rc"] == 17, r     # ERR_INVALID_EXT_SAMPLE (wire.h:152)
