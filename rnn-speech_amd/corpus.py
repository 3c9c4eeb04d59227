"""Corpus discovery for the training / test directories named in config.ini.

Host-side counterpart of the reference's util/dataprocessor.py:22-67 (constructor: detect the corpus
type of every directory, list (audio, transcript) pairs, attach durations, cache the list, drop short
items) and :208-337 (the four directory layouts).  Same results, different machinery:

  * one `os.walk` per directory feeds an index {suffix -> files}; the reference walks the tree once per
    probe and once per scanner;
  * durations come from the container headers (RIFF/WAVE `fmt `+`data`, FLAC STREAMINFO, NIST SPHERE,
    Ogg granule position) read directly -- the reference asks `mutagen`, which this image does not have;
  * TED-LIUM segments are cut out of the .sph files natively (16-bit PCM SPHERE) instead of shelling out
    to `sox` (:331-337); the segment files keep the reference's names (`<talk>_<start>.wav`) so a tree
    prepared by the reference is reused as is;
  * the file-list cache keeps the reference's pickle layout `[raw_data_paths, data]` (:251-261), so cache
    files are interchangeable.

Items are `[audio_path, cleaned_transcript, duration_seconds]` exactly as the reference's.
"""
import configparser
import logging
import os
import pickle
import struct
import wave
from concurrent.futures import ThreadPoolExecutor

from . import labels as _labels

DEFAULT_MIN_TEXT_LENGTH = 3      # characters  (reference util/dataprocessor.py:16)
DEFAULT_MIN_AUDIO_LENGTH = 0.4   # seconds     (reference util/dataprocessor.py:17)

# probe order of the reference's get_type (:208-226): first marker found wins
_MARKERS = ((".trn", "Vystadial_2013"), (".stm", "TEDLIUM"), ("index.tags.txt", "Shtooka"),
            (".trans.txt", "LibriSpeech"))


# ----------------------------------------------------------------------------- file index
def find_files(root_search_path, files_extension):
    """All files under the root whose name ends with the extension (reference :228-233)."""
    hits = []
    for root, _, files in os.walk(root_search_path):
        hits.extend(os.path.join(root, f) for f in files if f.endswith(files_extension))
    return hits


class _Tree(object):
    """One walk of a corpus directory, queried by suffix."""

    def __init__(self, root):
        self.root = root
        self.files = []
        for d, _, names in os.walk(root):
            self.files.extend(os.path.join(d, n) for n in names)

    def ending(self, suffix):
        return [f for f in self.files if f.endswith(suffix)]


def corpus_type(raw_data_path, tree=None):
    tree = tree or _Tree(raw_data_path)
    for suffix, name in _MARKERS:
        if tree.ending(suffix):
            return name
    return "Unrecognized"


# ----------------------------------------------------------------------------- durations
def _wav_duration(path):
    with wave.open(path, "rb") as w:
        return w.getnframes() / float(w.getframerate())


def _flac_duration(fh):
    # "fLaC", then metadata blocks; STREAMINFO (type 0) is mandatory and first: bytes 10..17 hold
    # sample rate (20 bits), channels-1 (3), bits-1 (5), total samples (36)
    hdr = fh.read(4)
    if len(hdr) < 4:
        return 0.0
    kind, size = hdr[0] & 0x7F, int.from_bytes(hdr[1:4], "big")
    body = fh.read(size)
    if kind != 0 or len(body) < 18:
        return 0.0
    packed = int.from_bytes(body[10:18], "big")
    rate = packed >> 44
    total = packed & ((1 << 36) - 1)
    return total / float(rate) if rate else 0.0


def sphere_header(path):
    """NIST SPHERE: 'NIST_1A\\n   1024\\n' then `key -type value` lines up to 'end_head'."""
    with open(path, "rb") as fh:
        if fh.read(8)[:7] != b"NIST_1A":
            raise ValueError("not a NIST SPHERE file: %s" % path)
        size = int(fh.read(8).strip())
        fh.seek(0)
        text = fh.read(size).decode("latin-1")
    fields = {"header_bytes": size}
    for line in text.split("\n")[2:]:
        parts = line.split(None, 2)
        if not parts or parts[0] == "end_head":
            break
        if len(parts) == 3:
            fields[parts[0]] = int(parts[2]) if parts[1] == "-i" else parts[2].strip()
    return fields


def _ogg_duration(path):
    # sample rate from the Vorbis / Opus identification header, length from the last page's granule
    with open(path, "rb") as fh:
        head = fh.read(4096)
        fh.seek(0, os.SEEK_END)
        size = fh.tell()
        fh.seek(max(0, size - 65536))
        tail = fh.read()
    rate = 0
    i = head.find(b"\x01vorbis")
    if i >= 0:
        rate = struct.unpack_from("<I", head, i + 12)[0]
    elif head.find(b"OpusHead") >= 0:
        rate = 48000
    j = tail.rfind(b"OggS")
    if rate == 0 or j < 0 or j + 14 > len(tail):
        return 0.0
    granule = struct.unpack_from("<q", tail, j + 6)[0]
    return max(granule, 0) / float(rate)


def audio_duration(path):
    """Seconds of audio in the file, 0 when the container is not recognised (the reference logs a
    warning and keeps 0 too, :235-244, so such files fall to the min_audio_size filter)."""
    try:
        with open(path, "rb") as fh:
            magic = fh.read(4)
            if magic == b"fLaC":
                return _flac_duration(fh)
        if magic == b"RIFF":
            return _wav_duration(path)
        if magic == b"NIST":
            h = sphere_header(path)
            return h.get("sample_count", 0) / float(h.get("sample_rate", 1))
        if magic == b"OggS":
            return _ogg_duration(path)
    except (OSError, ValueError, wave.Error, struct.error, EOFError) as exc:
        logging.warning("Audio file incorrect : %s (%s)", path, exc)
        return 0.0
    logging.warning("Audio file incorrect : %s", path)
    return 0.0


# ----------------------------------------------------------------------------- the four layouts
def _librispeech(tree):
    """<spk>/<chap>/<spk>-<chap>.trans.txt lines `<utt-id> TRANSCRIPT`, audio <utt-id>.flac beside it
    (reference :263-278: every .txt file is read, a file is abandoned at its first line whose head is
    shorter than 5 characters)."""
    for listing in tree.ending(".txt"):
        folder = os.path.dirname(listing)
        with open(listing, "r") as fh:
            for line in fh.read().split("\n"):
                utt = line.split(" ")[0]
                if len(utt) < 5:
                    break
                audio = folder + "/" + utt + ".flac"
                if os.path.exists(audio):
                    yield [audio, _labels.clean_label(line.replace(utt, "")), None]


def _shtooka(tree):
    """index.tags.txt is an ini file: one section per audio file, SWAC_TEXT = transcript (:280-293)."""
    for index in tree.ending("index.tags.txt"):
        ini = configparser.ConfigParser(comment_prefixes=("#", ";", "\\"))
        ini.read(index)
        base = index[:-len("index.tags.txt")]
        for name in ini.sections():
            if os.path.exists(base + name):
                yield [base + name, _labels.clean_label(ini[name]["SWAC_TEXT"]), None]


def _vystadial(tree):
    """x.wav with its transcript on the first line of x.wav.trn (:295-304)."""
    for audio in tree.ending(".wav"):
        if os.path.exists(audio + ".trn"):
            with open(audio + ".trn", "r") as fh:
                yield [audio, _labels.clean_label(fh.readline()), None]


def cut_sphere_segment(sph_file, wav_file, start, end):
    """Seconds [start, end) of a 16-bit PCM SPHERE file -> RIFF/WAVE (what the reference's
    `sox in.sph out.wav trim start =end` produces, :331-337).  False when the source is unusable."""
    try:
        h = sphere_header(sph_file)
        coding = str(h.get("sample_coding", "pcm"))
        width = int(h.get("sample_n_bytes", 2))
        if not coding.startswith("pcm") or "shorten" in coding or width != 2:
            logging.warning("SPHERE coding %s not supported: %s", coding, sph_file)
            return False
        rate, nch = int(h["sample_rate"]), int(h.get("channel_count", 1))
        first = max(0, int(round(float(start) * rate)))
        last = min(int(h.get("sample_count", 1 << 62)), int(round(float(end) * rate)))
        with open(sph_file, "rb") as fh:
            fh.seek(h["header_bytes"] + first * width * nch)
            pcm = fh.read(max(0, last - first) * width * nch)
        if str(h.get("sample_byte_format", "01")) == "10":          # big-endian samples
            swapped = bytearray(pcm)
            swapped[0::2], swapped[1::2] = pcm[1::2], pcm[0::2]
            pcm = bytes(swapped)
        with wave.open(wav_file, "wb") as w:
            w.setnchannels(nch)
            w.setsampwidth(width)
            w.setframerate(rate)
            w.writeframes(pcm)
        return True
    except (OSError, KeyError, ValueError) as exc:
        logging.warning("Execution failed : %s", exc)
        return False


def _tedlium(tree):
    """<set>/stm/<talk>.stm lines `talk chan speaker start end <labels> text`; audio ../sph/<talk>.sph,
    one wav per segment named <talk>_<start>.wav (:306-329)."""
    for stm in tree.ending(".stm"):
        folder = os.path.split(stm)[0]
        with open(stm, "r") as fh:
            for line in fh.read().split("\n"):
                if line == "":
                    continue
                f = line.split(" ", maxsplit=6)
                if len(f) < 7 or f[2] == "inter_segment_gap" or f[6] == "ignore_time_segment_in_scoring":
                    continue
                talk, start, end = f[0], f[3], f[4]
                wav = folder + "/../sph/{0}_{1}.wav".format(talk, start)
                if os.path.exists(wav) or cut_sphere_segment(folder + "/../sph/{0}.sph".format(talk), wav, start, end):
                    yield [wav, _labels.clean_label(f[6]), None]


_SCANNERS = {"LibriSpeech": _librispeech, "Shtooka": _shtooka, "Vystadial_2013": _vystadial, "TEDLIUM": _tedlium}


def scan(raw_data_path):
    tree = _Tree(raw_data_path)
    kind = corpus_type(raw_data_path, tree)
    if kind not in _SCANNERS:
        raise Exception("ERROR : unknown training_dataset_type")
    return list(_SCANNERS[kind](tree))


# ----------------------------------------------------------------------------- the reference's class
class DataProcessor(object):
    """`DataProcessor(dirs, file_cache).get_dataset()` of the reference (:21-71), plus the label codec
    statics (:73-205)."""

    clean_label = staticmethod(_labels.clean_label)
    get_str_labels = staticmethod(_labels.get_str_labels)
    get_labels_str = staticmethod(_labels.get_labels_str)
    get_str_to_one_hot_encoded = staticmethod(_labels.get_str_to_one_hot_encoded)
    find_files = staticmethod(find_files)
    extract_wav_from_sph = staticmethod(cut_sphere_segment)

    def __init__(self, raw_data_paths, file_cache=None, min_text_size=DEFAULT_MIN_TEXT_LENGTH,
                 min_audio_size=DEFAULT_MIN_AUDIO_LENGTH):
        self.raw_data_paths = raw_data_paths.replace(" ", "").split(",")
        self.file_cache = file_cache
        self.min_text_size = min_text_size
        self.min_audio_size = min_audio_size
        data = self.load_filelist()
        if data is not None:
            logging.info("%s : Using audio files list from cache file.", self.raw_data_paths)
        else:
            data = []
            for path in self.raw_data_paths:
                data += scan(path)
            logging.info("Retrieving audio duration from %d files. Please wait.", len(data))
            data = self._add_audio_length_on_dataset(data)
            if self.file_cache is not None:
                logging.info("%s : Saving audio files list to cache file.", self.raw_data_paths)
                self.save_filelist(data)
        if len(data) == 0:
            raise Exception("ERROR : no data found in directories {0}".format(self.raw_data_paths))
        self.data = [it for it in data if len(it[1]) > self.min_text_size and it[2] > self.min_audio_size]

    def get_dataset(self):
        return self.data

    @classmethod
    def get_type(cls, raw_data_path):
        return corpus_type(raw_data_path)

    @staticmethod
    def _add_audio_length_on_file(audio_file, text, _length):
        return [audio_file, text, audio_duration(audio_file)]

    @staticmethod
    def _add_audio_length_on_dataset(file_list):
        # header reads are I/O bound: threads, not the reference's process pool
        with ThreadPoolExecutor(max_workers=min(32, (os.cpu_count() or 1) * 2)) as pool:
            return list(pool.map(lambda it: DataProcessor._add_audio_length_on_file(*it), file_list))

    def save_filelist(self, data):
        with open(self.file_cache, "wb") as fh:
            pickle.dump([self.raw_data_paths, data], fh)

    def load_filelist(self):
        if self.file_cache is not None and os.path.exists(self.file_cache):
            with open(self.file_cache, "rb") as fh:
                paths, data = pickle.load(fh)
            if paths == self.raw_data_paths:
                return data
        return None
