"""Frames (+ audio) -> .mp4: the output path's last stage.

The reference muxes with OpenCV + moviepy (/root/reference/*phoneme_data/VidTIMIT/fadg0/image2video_real.py:17-38;
`python image2video_real_audio_text2video.py "$1" $2`, text2video_audio.sh:44): cv2.VideoWriter('MP4V', 25 fps) over
the sorted fake_B_*.jpg frames, then moviepy's write_videofile(audio=<mp3>) adds the sound track (an ffmpeg
re-encode).  This image has neither OpenCV nor ffmpeg nor any other video encoder, and the frame loop already holds
every frame as a JPEG, so the mux here is a pure container operation: an ISO base-media (.mp4) file whose video track
carries the JPEG frames as they are (sample entry 'mp4v' with objectTypeIndication 0x6C = ISO/IEC 10918-1 JPEG --
how ffmpeg itself stores `-c:v mjpeg` in .mp4) at the reference's 25 frames/s, plus an optional audio track taken
without re-encoding from a .wav (16-bit PCM, sample entry 'sowt') or an .mp3 (one sample per MPEG audio frame, 'mp4a'
with objectTypeIndication 0x6B / 0x69).  No frame is decoded or re-compressed: the video is bit-identical to the
frames on disk.  When `cv2` and `moviepy` ARE importable the reference's exact route is taken instead
(`reference_route=True`), and a request for any other codec fails loudly.
"""
import glob
import os
import struct
import wave

FPS = 25  # image2video_real.py:12


def _box(kind, *payload):
    data = b"".join(payload)
    return struct.pack(">I4s", 8 + len(data), kind) + data


def _full(kind, version, flags, *payload):
    return _box(kind, struct.pack(">I", (version << 24) | flags), *payload)


def _descr(tag, payload):
    n = len(payload)
    assert n < (1 << 21)
    return bytes([tag, 0x80 | (n >> 14) & 0x7f, 0x80 | (n >> 7) & 0x7f, n & 0x7f]) + payload


def _esds(object_type, stream_type, bitrate=0):
    dec = struct.pack(">BB", object_type, (stream_type << 2) | 1) + b"\x00\x00\x00" + struct.pack(">II", bitrate, bitrate)
    es = struct.pack(">HB", 0, 0) + _descr(4, dec) + _descr(6, b"\x02")
    return _full(b"esds", 0, 0, _descr(3, es))


MATRIX = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def jpeg_size(data):
    """(width, height) from the SOF marker of a JPEG stream"""
    i = 2
    while i + 9 < len(data):
        if data[i] != 0xFF:
            raise ValueError("not a JPEG stream")
        m = data[i + 1]
        if m in (0xC0, 0xC1, 0xC2):
            h, w = struct.unpack(">HH", data[i + 5:i + 9])
            return w, h
        i += 2 + struct.unpack(">H", data[i + 2:i + 4])[0]
    raise ValueError("JPEG without SOF marker")


# MPEG audio frame header tables (ISO/IEC 11172-3 / 13818-3), layer III
_BITRATE = {1: [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320],
            2: [0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160]}
_RATE = {3: [44100, 48000, 32000], 2: [22050, 24000, 16000], 0: [11025, 12000, 8000]}


def mp3_frames(data):
    """[(offset, size)], sample rate, channels, samples per frame, MPEG version id of an .mp3 byte string"""
    i = 0
    if data[:3] == b"ID3":
        n = (data[6] & 0x7f) << 21 | (data[7] & 0x7f) << 14 | (data[8] & 0x7f) << 7 | (data[9] & 0x7f)
        i = 10 + n
    frames, rate, ch, spf, ver = [], None, None, None, None
    while i + 4 <= len(data):
        h = struct.unpack(">I", data[i:i + 4])[0]
        if (h >> 21) != 0x7FF:
            i += 1
            continue
        v, layer, br, sr, pad = (h >> 19) & 3, (h >> 17) & 3, (h >> 12) & 15, (h >> 10) & 3, (h >> 9) & 1
        if v == 1 or layer != 1 or br in (0, 15) or sr == 3:
            i += 1
            continue
        r = _RATE[v][sr]
        kb = _BITRATE[1 if v == 3 else 2][br]
        size = (144 if v == 3 else 72) * kb * 1000 // r + pad
        if rate is None:
            rate, ch, spf, ver = r, 1 if ((h >> 6) & 3) == 3 else 2, 1152 if v == 3 else 576, v
        if r != rate or i + size > len(data):
            break
        frames.append((i, size))
        i += size
    if not frames:
        raise ValueError("no MPEG layer-III frames found")
    return frames, rate, ch, spf, ver


def _trak(track_id, handler, name, timescale, duration, movie_duration, media_header, sample_entry, stts, sizes,
          chunk_offset, width=0, height=0, volume=0):
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, track_id, 0, movie_duration), b"\x00" * 8,
                 struct.pack(">hhhH", 0, 0, volume, 0), MATRIX, struct.pack(">II", width << 16, height << 16))
    mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, timescale, duration, 0x55C4, 0))
    hdlr = _full(b"hdlr", 0, 0, struct.pack(">I4s", 0, handler), b"\x00" * 12, name + b"\x00")
    dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1), _full(b"url ", 0, 1)))
    if isinstance(sizes, int):      # constant sample size
        stsz = _full(b"stsz", 0, 0, struct.pack(">II", sizes, stts[0][0]))
        n_samples = stts[0][0]
    else:
        stsz = _full(b"stsz", 0, 0, struct.pack(">II", 0, len(sizes)), struct.pack(">%dI" % len(sizes), *sizes))
        n_samples = len(sizes)
    stbl = _box(b"stbl",
                _full(b"stsd", 0, 0, struct.pack(">I", 1), sample_entry),
                _full(b"stts", 0, 0, struct.pack(">I", len(stts)), b"".join(struct.pack(">II", c, d) for c, d in stts)),
                _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n_samples, 1)),     # ONE chunk holding every sample
                stsz,
                _full(b"co64", 0, 0, struct.pack(">IQ", 1, chunk_offset)))
    return _box(b"trak", tkhd, _box(b"mdia", mdhd, hdlr, _box(b"minf", media_header, dinf, stbl)))


def write_mp4(frames, out_path, fps=FPS, audio=None):
    """frames: JPEG files (paths) or JPEG byte strings, in display order; audio: None | path to .wav (16-bit PCM)
    | .mp3.  Writes out_path and returns {"frames", "width", "height", "seconds", "audio"}."""
    blobs = []
    for f in frames:
        if isinstance(f, (bytes, bytearray)):
            blobs.append(bytes(f))
        else:
            with open(f, "rb") as fh:
                blobs.append(fh.read())
    if not blobs:
        raise ValueError("write_mp4: no frames")
    w, h = jpeg_size(blobs[0])
    for b in blobs:
        if jpeg_size(b) != (w, h):
            raise ValueError("write_mp4: frames of different sizes")
    vscale, vdelta = fps * 512, 512
    vdur = len(blobs) * vdelta
    a = None
    if audio is not None:
        ext = os.path.splitext(audio)[1].lower()
        if ext == ".wav":
            with wave.open(audio, "rb") as wf:
                if wf.getsampwidth() != 2 or wf.getcomptype() != "NONE":
                    raise ValueError("write_mp4: only 16-bit PCM .wav audio is supported, got %d-bit %s"
                                     % (8 * wf.getsampwidth(), wf.getcomptype()))
                ch, rate, n = wf.getnchannels(), wf.getframerate(), wf.getnframes()
                pcm = wf.readframes(n)
            entry = _box(b"sowt", b"\x00" * 6, struct.pack(">H", 1), struct.pack(">HHIHHHHI", 0, 0, 0, ch, 16, 0, 0, rate << 16))
            a = dict(data=pcm, scale=rate, dur=n, entry=entry, stts=[(n, 1)], sizes=2 * ch, kind="pcm_s16le")
        elif ext == ".mp3":
            with open(audio, "rb") as fh:
                raw = fh.read()
            fr, rate, ch, spf, ver = mp3_frames(raw)
            data = b"".join(raw[o:o + s] for o, s in fr)
            entry = _box(b"mp4a", b"\x00" * 6, struct.pack(">H", 1), struct.pack(">HHIHHHHI", 0, 0, 0, ch, 16, 0, 0, rate << 16),
                         _esds(0x6B if ver == 3 else 0x69, 0x05, 8 * len(data) * rate // (len(fr) * spf)))
            a = dict(data=data, scale=rate, dur=len(fr) * spf, entry=entry, stts=[(len(fr), spf)], sizes=[s for _, s in fr],
                     kind="mp3")
        else:
            raise ValueError("write_mp4: audio must be .wav (16-bit PCM) or .mp3, got %r" % audio)
    mscale = 1000
    mdur = vdur * mscale // vscale
    if a:
        mdur = max(mdur, a["dur"] * mscale // a["scale"])
    ftyp = _box(b"ftyp", b"isom", struct.pack(">I", 0x200), b"isomiso2mp41")
    video = b"".join(blobs)
    v_off = len(ftyp) + 16                      # mdat with a 64-bit size field
    a_off = v_off + len(video)
    mdat = struct.pack(">I4sQ", 1, b"mdat", 16 + len(video) + (len(a["data"]) if a else 0))
    ventry = _box(b"mp4v", b"\x00" * 6, struct.pack(">H", 1), b"\x00" * 16, struct.pack(">HHIIIH", w, h, 0x480000, 0x480000, 0, 1),
                  b"\x00" * 32, struct.pack(">Hh", 0x18, -1), _esds(0x6C, 0x04, 8 * len(video) * fps // len(blobs)))
    traks = [_trak(1, b"vide", b"VideoHandler", vscale, vdur, vdur * mscale // vscale, _full(b"vmhd", 0, 1, b"\x00" * 8), ventry,
                   [(len(blobs), vdelta)], [len(b) for b in blobs], v_off, w, h)]
    if a:
        traks.append(_trak(2, b"soun", b"SoundHandler", a["scale"], a["dur"], a["dur"] * mscale // a["scale"],
                           _full(b"smhd", 0, 0, b"\x00" * 4), a["entry"], a["stts"], a["sizes"], a_off, volume=0x100))
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIIIIH", 0, 0, mscale, mdur, 0x10000, 0x100), b"\x00" * 10, MATRIX, b"\x00" * 24,
                 struct.pack(">I", len(traks) + 1))
    with open(out_path, "wb") as fh:
        fh.write(ftyp)
        fh.write(mdat)
        fh.write(video)
        if a:
            fh.write(a["data"])
        fh.write(_box(b"moov", mvhd, *traks))
    return {"frames": len(blobs), "width": w, "height": h, "seconds": len(blobs) / float(fps), "audio": a["kind"] if a else None}


def _reference_route(frames, out_path, fps, audio):
    """the reference's own stages, when its two libraries exist (image2video_real.py:17-38)"""
    import cv2
    import moviepy.editor as mpe
    tmp = out_path + ".noaudio.mp4"
    out = None
    for f in frames:
        img = cv2.imread(f)
        if out is None:
            out = cv2.VideoWriter(tmp, cv2.VideoWriter_fourcc(*"MP4V"), fps, (img.shape[1], img.shape[0]))
        out.write(img)
    if out is not None:
        out.release()
    clip = mpe.VideoFileClip(tmp)
    clip.write_videofile(out_path, audio=audio)
    os.remove(tmp)


def image2video(text, person, results_dir="./results", audio_dirs=("../Text2Video/input_audio_real", "../Text2Video/input_audio"),
                tests=("tmp", "tmp_smooth"), fps=FPS, reference_route=None):
    """What `python image2video_real_audio_text2video.py "$1" $2` does after test.py (text2video_audio.sh:44), run from
    the vid2vid directory: results/<person>/test_latest/<test>/fake_B_*.jpg (sorted) -> results/<person>/<person>_<test>.mp4
    at 25 fps with the utterance's audio (file name = the first 10 characters of the text without blanks /
    punctuation, as interp_landmarks_motion_phoneme_VidTIMIT_smooth.py:20-24 derives it).  The un-vendored script's
    exact paths are [RECALL]; the pattern and output name follow the commented lines of image2video_real.py:14-15,36."""
    import re
    import string
    stem = re.sub(r"[%s]+" % re.escape(string.punctuation + "，。！？、；：“”‘’（）《》"), "", re.sub(" ", "", text))[:10]
    audio = None
    for d in audio_dirs:
        for ext in (".mp3", ".wav"):
            p = os.path.join(d, person, stem + ext)
            if audio is None and os.path.exists(p):
                audio = p
    if reference_route is None:
        try:
            import cv2  # noqa: F401
            import moviepy.editor  # noqa: F401
            reference_route = True
        except ImportError:
            reference_route = False
    done = []
    for test in tests:
        frames = sorted(glob.glob(os.path.join(results_dir, person, "test_latest", test, "fake_B_*.jpg")))
        if not frames:
            continue
        out = os.path.join(results_dir, person, "%s_%s.mp4" % (person, test))
        if reference_route:
            _reference_route(frames, out, fps, audio)
            done.append((out, {"frames": len(frames), "audio": audio, "codec": "MP4V + moviepy (reference route)"}))
        else:
            info = write_mp4(frames, out, fps, audio)
            info["codec"] = "JPEG frames in mp4 (no video encoder in this image)"
            done.append((out, info))
    if not done:
        raise FileNotFoundError("image2video: no fake_B_*.jpg under %s" % os.path.join(results_dir, person, "test_latest"))
    return done
