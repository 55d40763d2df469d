"""The output path's mux stage: frames (+ audio) -> .mp4 without a video encoder.  The container is
parsed back here box by box: the video samples must be the frame files bit for bit, at the reference's 25 fps, and
the audio samples the .wav payload / the .mp3's MPEG frames."""
import io
import os
import struct
import wave

import numpy as np
import pytest
from PIL import Image

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _boxes(buf, start=0, end=None):
    end = len(buf) if end is None else end
    out, i = [], start
    while i + 8 <= end:
        size, kind = struct.unpack(">I4s", buf[i:i + 8])
        hdr = 8
        if size == 1:
            size = struct.unpack(">Q", buf[i + 8:i + 16])[0]
            hdr = 16
        out.append((kind, i + hdr, i + size))
        i += size
    return out


def _find(buf, path, start=0, end=None):
    for kind, s, e in _boxes(buf, start, end):
        if kind == path[0]:
            return (s, e) if len(path) == 1 else _find(buf, path[1:], s, e)
    raise KeyError(path)


def _track_samples(buf, trak):
    s, e = trak
    stbl = _find(buf, [b"mdia", b"minf", b"stbl"], s, e)
    stsz = _find(buf, [b"stsz"], *stbl)
    fixed, count = struct.unpack(">II", buf[stsz[0] + 4:stsz[0] + 12])
    sizes = [fixed] * count if fixed else list(struct.unpack(">%dI" % count, buf[stsz[0] + 12:stsz[0] + 12 + 4 * count]))
    co = _find(buf, [b"co64"], *stbl)
    n, off = struct.unpack(">IQ", buf[co[0] + 4:co[0] + 16])
    assert n == 1
    stts = _find(buf, [b"stts"], *stbl)
    n_e = struct.unpack(">I", buf[stts[0] + 4:stts[0] + 8])[0]
    entries = [struct.unpack(">II", buf[stts[0] + 8 + 8 * k:stts[0] + 16 + 8 * k]) for k in range(n_e)]
    mdhd = _find(buf, [b"mdia", b"mdhd"], s, e)
    timescale, duration = struct.unpack(">II", buf[mdhd[0] + 12:mdhd[0] + 20])
    stsd = _find(buf, [b"stsd"], *stbl)
    fourcc = buf[stsd[0] + 12:stsd[0] + 16]
    samples = []
    for sz in sizes:
        samples.append(buf[off:off + sz])
        off += sz
    return {"samples": samples, "stts": entries, "timescale": timescale, "duration": duration, "fourcc": fourcc}


def _frames(tmp, n, size=(96, 64)):
    rng = np.random.default_rng(0)
    paths = []
    for i in range(n):
        a = (rng.random((size[1], size[0], 3)) * 255).astype(np.uint8)
        a[:, : 4 * i + 4] = (10 * i) % 255
        p = os.path.join(tmp, "fake_B_%04d.jpg" % i)
        Image.fromarray(a).save(p, quality=90)
        paths.append(p)
    return paths


@pytest.mark.parametrize("audio", [None, "wav", "mp3"])
def test_mp4_roundtrip(tmp_path, audio):
    from text2video_amd import mux
    frames = _frames(str(tmp_path), 7)
    apath = None
    if audio == "wav":
        apath = str(tmp_path / "a.wav")
        pcm = (np.sin(np.arange(16000 // 2) * 0.05) * 12000).astype("<i2")
        with wave.open(apath, "wb") as wf:
            wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000); wf.writeframes(pcm.tobytes())
    elif audio == "mp3":
        apath = os.path.join(GOLD, "audio", "Shehadyour.mp3")     # the reference's own utterance audio (input_audio_real/fadg0)
    out = str(tmp_path / "o.mp4")
    info = mux.write_mp4(frames, out, audio=apath)
    assert info["frames"] == 7 and (info["width"], info["height"]) == (96, 64) and abs(info["seconds"] - 7 / 25.0) < 1e-9
    buf = open(out, "rb").read()
    top = _boxes(buf)
    assert [k for k, _, _ in top] == [b"ftyp", b"mdat", b"moov"] and top[-1][2] == len(buf)
    moov = _find(buf, [b"moov"])
    traks = [(s, e) for k, s, e in _boxes(buf, *moov) if k == b"trak"]
    assert len(traks) == (2 if audio else 1)
    v = _track_samples(buf, traks[0])
    assert v["fourcc"] == b"mp4v" and v["stts"] == [(7, 512)] and v["timescale"] == 25 * 512 and v["duration"] == 7 * 512
    for s, p in zip(v["samples"], frames):
        assert s == open(p, "rb").read()                              # the frames, bit for bit
        assert Image.open(io.BytesIO(s)).size == (96, 64)
    if audio == "wav":
        a = _track_samples(buf, traks[1])
        assert a["fourcc"] == b"sowt" and a["timescale"] == 16000 and a["duration"] == 8000 and a["stts"] == [(8000, 1)]
        assert b"".join(a["samples"]) == pcm.tobytes()
    if audio == "mp3":
        a = _track_samples(buf, traks[1])
        raw = open(apath, "rb").read()
        fr, rate, ch, spf, ver = mux.mp3_frames(raw)
        assert a["fourcc"] == b"mp4a" and a["timescale"] == rate == 32000 and a["stts"] == [(len(fr), 1152)]
        assert a["samples"] == [raw[o:o + s] for o, s in fr] and len(fr) == 135
        assert all(s[0] == 0xFF and (s[1] & 0xE0) == 0xE0 for s in a["samples"])    # every sample starts on a frame sync


def test_image2video_drop_in_layout_and_errors(tmp_path, monkeypatch):
    """`image2video "<text>" <person>` from the vid2vid directory: the reference's frame pattern and output names."""
    from text2video_amd import mux
    work = tmp_path / "vid2vid"
    for test in ("tmp", "tmp_smooth"):
        d = work / "results" / "fadg0" / "test_latest" / test
        d.mkdir(parents=True)
        _frames(str(d), 4)
    adir = tmp_path / "Text2Video" / "input_audio_real" / "fadg0"
    adir.mkdir(parents=True)
    import shutil
    shutil.copy(os.path.join(GOLD, "audio", "Shehadyour.mp3"), str(adir / "Shehadyour.mp3"))
    monkeypatch.chdir(work)
    done = mux.image2video("She had your dark suit in greasy wash water all year.", "fadg0", reference_route=False)
    assert [os.path.relpath(p) for p, _ in done] == ["results/fadg0/fadg0_tmp.mp4", "results/fadg0/fadg0_tmp_smooth.mp4"]
    assert all(i["frames"] == 4 and i["audio"] == "mp3" for _, i in done)
    with pytest.raises(FileNotFoundError):
        mux.image2video("x", "nobody", reference_route=False)
    with pytest.raises(ValueError, match="audio must be"):
        mux.write_mp4(_frames(str(tmp_path), 2), str(tmp_path / "x.mp4"), audio="a.ogg")
    with pytest.raises(ValueError, match="no frames"):
        mux.write_mp4([], str(tmp_path / "x.mp4"))
