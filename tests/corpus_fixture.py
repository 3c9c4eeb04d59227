"""Synthetic corpus trees in the reference's four directory layouts (shared by tests/test_cpu_host.py and
tools/make_golden.py, which runs the reference's own DataProcessor over the same trees)."""
import os
import wave

import numpy as np


def write_wav(path, seconds, rate=16000):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes((np.zeros(int(seconds * rate), np.int16)).tobytes())


def write_flac_header(path, total_samples, rate=16000):
    """"fLaC" + a last-metadata-block STREAMINFO (34 bytes): enough for duration probes."""
    info = bytearray(34)
    info[0:2] = (4096).to_bytes(2, "big")
    info[2:4] = (4096).to_bytes(2, "big")
    packed = (rate << 44) | (0 << 41) | (15 << 36) | total_samples      # 1 channel, 16 bits
    info[10:18] = packed.to_bytes(8, "big")
    with open(path, "wb") as fh:
        fh.write(b"fLaC" + bytes([0x80]) + (34).to_bytes(3, "big") + bytes(info))


def write_sphere(path, samples, rate=16000, byte_format="01"):
    head = ("NIST_1A\n   1024\nsample_count -i %d\nsample_rate -i %d\nchannel_count -i 1\nsample_n_bytes -i 2\n"
            "sample_byte_format -s2 %s\nsample_coding -s3 pcm\nend_head\n" % (len(samples), rate, byte_format))
    raw = samples.astype("<i2" if byte_format == "01" else ">i2").tobytes()
    with open(path, "wb") as fh:
        fh.write(head.encode().ljust(1024, b" ") + raw)


def build_trees(root, ted_segments=False):
    """Creates root/{libri,vystadial,shtooka,ted}; returns the four directory paths.  With
    `ted_segments` the TED-LIUM segment wavs already exist (the reference needs `sox` to cut them)."""
    root = str(root)
    ls = os.path.join(root, "libri", "19", "198")
    os.makedirs(ls)
    with open(os.path.join(ls, "19-198.trans.txt"), "w") as fh:
        fh.write("19-198-0000 NORTHANGER ABBEY\n19-198-0001 THIS LITTLE WORK WAS FINISHED IN THE YEAR 1803\n"
                 "19-198-0002 MISSING AUDIO\n19-198-0003 SHORT ONE\n19-198-0004 IT'S MR. O'BRIEN'S\n")
    write_flac_header(os.path.join(ls, "19-198-0000.flac"), 16000 * 3)
    write_flac_header(os.path.join(ls, "19-198-0001.flac"), 16000 * 5)
    write_flac_header(os.path.join(ls, "19-198-0003.flac"), 1600)            # 0.1 s: below min_audio_size
    write_flac_header(os.path.join(ls, "19-198-0004.flac"), 16000 * 2)
    ls2 = os.path.join(root, "libri", "26", "495")
    os.makedirs(ls2)
    with open(os.path.join(ls2, "26-495.trans.txt"), "w") as fh:
        fh.write("26-495-0000 A SECOND CHAPTER\n\n26-495-0001 AFTER THE BLANK LINE\n")
    write_flac_header(os.path.join(ls2, "26-495-0000.flac"), 16000 * 4)
    write_flac_header(os.path.join(ls2, "26-495-0001.flac"), 16000 * 4)

    vy = os.path.join(root, "vystadial")
    os.makedirs(vy)
    write_wav(os.path.join(vy, "a.wav"), 1.0)
    with open(os.path.join(vy, "a.wav.trn"), "w") as fh:
        fh.write("HELLO THERE (NOISE) friend\nsecond line ignored\n")
    write_wav(os.path.join(vy, "b.wav"), 1.0)                                # no transcript -> skipped
    write_wav(os.path.join(vy, "c.wav"), 2.5)
    with open(os.path.join(vy, "c.wav.trn"), "w") as fh:
        fh.write("ok\n")                                                      # text too short -> filtered

    sh = os.path.join(root, "shtooka", "flac")
    os.makedirs(sh)
    write_flac_header(os.path.join(sh, "eng - apple.flac"), 16000)
    with open(os.path.join(sh, "index.tags.txt"), "w") as fh:
        fh.write("# comment\n[eng - apple.flac]\nSWAC_TEXT=an apple\nSWAC_LANG=eng\n"
                 "[eng - gone.flac]\nSWAC_TEXT=missing file\n")

    ted = os.path.join(root, "ted", "train")
    os.makedirs(os.path.join(ted, "stm"))
    os.makedirs(os.path.join(ted, "sph"))
    ramp = (np.arange(16000 * 4) % 30000).astype(np.int16)
    write_sphere(os.path.join(ted, "sph", "TalkA.sph"), ramp, byte_format="10")
    with open(os.path.join(ted, "stm", "TalkA.stm"), "w") as fh:
        fh.write("TalkA 1 inter_segment_gap 0 0.5 <o,,unknown> ignore_time_segment_in_scoring\n"
                 "TalkA 1 Speaker 0.5 2.0 <o,f0,female> the first segment of speech\n"
                 "TalkA 1 Speaker 2.0 3.25 <o,f0,female> ignore_time_segment_in_scoring\n"
                 "TalkA 1 Speaker 3.0 3.9 <o,f0,female> and the last one\n")
    if ted_segments:
        write_wav(os.path.join(ted, "sph", "TalkA_0.5.wav"), 1.5)
        write_wav(os.path.join(ted, "sph", "TalkA_3.0.wav"), 0.9)
    return [os.path.join(root, d) for d in ("libri", "vystadial", "shtooka", "ted")], ramp
