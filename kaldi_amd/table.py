"""Kaldi's table / command-line layer, as far as the decode binaries use it:

  * ParseOptions (util/parse-options.cc:309-600): --name=value options with name normalisation,
    bare --flag for booleans, --config=file, "--" ends the options, prefixed registration;
  * rspecifiers / wspecifiers / extended filenames (util/kaldi-table.cc:115-310,
    util/kaldi-io.cc:85-186) -- classification is done by the C library (table.cc);
  * SequentialTableReader / RandomAccessTableReader / TableWriter over archives, script files,
    pipes and standard input for the object types on the decode path: float matrices
    (features, log-likelihoods, ivectors), int32 vectors, waveforms, lattices.

Host plumbing only: nothing here computes."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

from . import io as kio
from ._lib import KamdError, check, lib

RX_NONE, RX_FILE, RX_STDIN, RX_OFFSET_FILE, RX_PIPE = range(5)
WX_NONE, WX_FILE, WX_STDOUT, WX_PIPE = range(4)
NO_SPECIFIER, ARCHIVE, SCRIPT, BOTH = range(4)


def classify_rxfilename(name):
    return lib().kamd_classify_rxfilename(name.encode())


def classify_wxfilename(name):
    return lib().kamd_classify_wxfilename(name.encode())


def classify_rspecifier(rspecifier):
    """-> (type, rxfilename, {'once','sorted','called_sorted','permissive','background'})"""
    buf = C.create_string_buffer(len(rspecifier) + 2)
    o = C.c_int()
    t = lib().kamd_classify_rspecifier(rspecifier.encode(), buf, len(rspecifier) + 2, C.byref(o))
    names = ("once", "sorted", "called_sorted", "permissive", "background")
    return t, buf.value.decode(), {n: bool(o.value >> i & 1) for i, n in enumerate(names)}


def classify_wspecifier(wspecifier):
    """-> (type, archive_wxfilename, script_wxfilename, {'binary','flush','permissive'})"""
    n = len(wspecifier) + 2
    a, s = C.create_string_buffer(n), C.create_string_buffer(n)
    o = C.c_int()
    t = lib().kamd_classify_wspecifier(wspecifier.encode(), a, n, s, n, C.byref(o))
    return t, a.value.decode(), s.value.decode(), {k: bool(o.value >> i & 1) for i, k in enumerate(("binary", "flush", "permissive"))}


class Input:
    """`with Input(rxfilename) as (path, offset):` -- a seekable file for any rxfilename; pipe and
    stdin contents are spooled to a temporary file that is removed on exit."""

    def __init__(self, rxfilename):
        self.rx = rxfilename

    def __enter__(self):
        buf = C.create_string_buffer(max(len(self.rx) + 2, 4096))
        off, tmp = C.c_int64(), C.c_int()
        check(lib().kamd_rx_materialize(self.rx.encode(), buf, len(buf), C.byref(off), C.byref(tmp)))
        self.path, self.temp = buf.value.decode(), bool(tmp.value)
        return self.path, off.value

    def __exit__(self, *exc):
        if self.temp:
            try:
                os.unlink(self.path)
            except OSError:
                pass
        return False


def read_script_file(rxfilename):
    """ReadScriptFile (util/kaldi-table.cc:56-84): [(key, rest of line)]; empty lines or lines
    with no second field are errors."""
    out = []
    with Input(rxfilename) as (path, off):
        with open(path, "rb") as f:
            f.seek(off)
            data = f.read()
    if b"\0" in data[:2]:
        raise KamdError("script file appears to be binary: " + rxfilename)
    for n, line in enumerate(data.decode().split("\n")[:-1] if data.endswith(b"\n") else data.decode().split("\n")):
        if line == "":
            raise KamdError("Empty %d'th line in script file %s" % (n + 1, rxfilename))
        parts = line.split(None, 1) if not line[0].isspace() else []
        if len(parts) != 2 or not parts[1].strip():
            raise KamdError("Invalid %d'th line in script file %s:\"%s\"" % (n + 1, rxfilename, line))
        out.append((parts[0], parts[1].rstrip("\r")))
    return out


# ---- holders: read one object at (path, offset[, key]) ------------------------------------
def _read_matrix_at(path, off):
    r, c = C.c_int32(), C.c_int32()
    o = C.c_int64(off)
    p = C.POINTER(C.c_float)()
    check(lib().kamd_ark_read_matrix(path.encode(), C.byref(o), None, 0, C.byref(r), C.byref(c), C.byref(p)))
    try:
        n = r.value * c.value
        return np.ctypeslib.as_array(p, (max(n, 1),)).copy()[:n].reshape(r.value, c.value)
    finally:
        lib().kamd_host_free(C.cast(p, C.c_void_p))


def _read_int32_at(path, off):
    n = C.c_int32()
    o = C.c_int64(off)
    p = C.POINTER(C.c_int32)()
    check(lib().kamd_ark_read_int32_vector(path.encode(), C.byref(o), None, 0, C.byref(n), C.byref(p)))
    try:
        return np.ctypeslib.as_array(p, (max(n.value, 1),)).copy()[:n.value]
    finally:
        lib().kamd_host_free(C.cast(p, C.c_void_p))


def _read_wave_at(path, off):
    if off:
        raise KamdError("waveforms inside archives (offset %d of %s) are not supported; list the files or commands in the scp" % (off, path))
    return kio.read_wave(path)


def _parse_vector(buf, pos):
    """Vector<float>::Read (matrix/kaldi-vector.cc:1113-1230) at buf[pos:]: binary FV / DV or text " [ 1 2 3 ]".
    -> (vector, position after it)"""
    import struct
    if buf[pos:pos + 2] == b"\0B":
        tok = buf[pos + 2:pos + 5]
        if tok not in (b"FV ", b"DV ") or buf[pos + 5] != 4:
            raise KamdError("vector expected, got %r" % tok)
        n = struct.unpack_from("<i", buf, pos + 6)[0]
        w = 4 if tok == b"FV " else 8
        v = np.frombuffer(buf, "<f4" if w == 4 else "<f8", n, pos + 10).astype(np.float32)
        return v, pos + 10 + w * n
    e = buf.index(b"]", pos)
    txt = buf[pos:e].decode().replace("[", " ").split()
    nl = buf.find(b"\n", e)
    return np.asarray([float(x) for x in txt], np.float32), (nl + 1 if nl >= 0 else len(buf))


def _parse_dmatrix(buf, pos):
    """Matrix<double>::Read (matrix/kaldi-matrix.cc:1393-1580) at buf[pos:]: binary DM / FM (converted) or the text form
    " [\n  1 2 \n  3 4 ]".  CMVN statistics travel as double matrices (DoubleMatrixWriter, compute-cmvn-stats.cc:93).
    -> (float64 matrix, position after it)"""
    import struct
    if buf[pos:pos + 2] == b"\0B":
        tok = buf[pos + 2:pos + 5]
        if tok not in (b"FM ", b"DM ") or buf[pos + 5] != 4 or buf[pos + 10] != 4:
            raise KamdError("matrix expected, got %r" % tok)
        r, c = struct.unpack_from("<i", buf, pos + 6)[0], struct.unpack_from("<i", buf, pos + 11)[0]
        w = 4 if tok == b"FM " else 8
        m = np.frombuffer(buf, "<f4" if w == 4 else "<f8", r * c, pos + 15).astype(np.float64).reshape(r, c)
        return m, pos + 15 + w * r * c
    e = buf.index(b"]", pos)
    rows = [[float(x) for x in ln.split()] for ln in buf[pos:e].decode().replace("[", " ").split("\n")]
    rows = [r for r in rows if r]
    if any(len(r) != len(rows[0]) for r in rows):
        raise KamdError("text matrix with rows of different lengths")
    nl = buf.find(b"\n", e)
    return np.asarray(rows, np.float64).reshape(len(rows), len(rows[0]) if rows else 0), (nl + 1 if nl >= 0 else len(buf))


def _read_dmatrix_at(path, off):
    with open(path, "rb") as f:
        f.seek(off)
        return _parse_dmatrix(f.read(), 0)[0]


def _read_dmatrix_ark(path):
    buf = open(path, "rb").read()
    pos = 0
    while True:
        while pos < len(buf) and buf[pos:pos + 1] in (b" ", b"\n"):
            pos += 1
        if pos >= len(buf):
            return
        e = buf.index(b" ", pos)
        key = buf[pos:e].decode()
        m, pos = _parse_dmatrix(buf, e + 1)
        yield key, m


def _dmatrix_bytes(m, binary):
    """Matrix<double>::Write (matrix/kaldi-matrix.cc:1355-1391)"""
    import struct
    m = np.ascontiguousarray(m, np.float64)
    if m.ndim != 2:
        raise KamdError("matrix expected")
    if binary:
        return b"\0BDM \4" + struct.pack("<i", m.shape[0]) + b"\4" + struct.pack("<i", m.shape[1]) + m.tobytes()
    if m.size == 0:
        return b" [ ]\n"
    return (" [" + "".join("\n  " + "".join("%.17g " % x for x in row) for row in m) + "]\n").encode()


def _read_vector_at(path, off):
    with open(path, "rb") as f:
        f.seek(off)
        return _parse_vector(f.read(), 0)[0]


def _read_vector_ark(path):
    buf = open(path, "rb").read()
    pos = 0
    while True:
        while pos < len(buf) and buf[pos:pos + 1] in (b" ", b"\n"):
            pos += 1
        if pos >= len(buf):
            return
        e = buf.index(b" ", pos)
        key = buf[pos:e].decode()
        v, pos = _parse_vector(buf, e + 1)
        yield key, v


def _read_token_ark(path):
    """TokenHolder / TokenVectorHolder text archives (utt2spk, spk2utt): key, then the tokens of the line"""
    for line in open(path):
        parts = line.split()
        if parts:
            yield parts[0], parts[1:]


_AT = {"matrix": _read_matrix_at, "int32": _read_int32_at, "wave": _read_wave_at, "vector": _read_vector_at, "dmatrix": _read_dmatrix_at}
_ARK = {"matrix": kio.read_matrix_ark, "int32": kio.read_int32_vector_ark, "vector": _read_vector_ark, "tokens": _read_token_ark,
        "dmatrix": _read_dmatrix_ark,
        "lattice": lambda path: ((k, (start, final, arcs)) for k, start, final, arcs in kio.read_lattices(path))}


class SequentialTableReader:
    """for key, value in SequentialTableReader("ark:feats.ark" | "scp:wav.scp" | "ark:gunzip -c x.gz |", kind)
    kind: "matrix" | "dmatrix" (double precision: CMVN statistics) | "vector" | "int32" | "wave" | "lattice" | "tokens" (text lines: utt2spk, spk2utt).  With the `p` (permissive) option scp entries
    that cannot be read are skipped, as the reference does; otherwise they raise."""

    def __init__(self, rspecifier, kind):
        self.type, self.rx, self.opts = classify_rspecifier(rspecifier)
        if self.type == NO_SPECIFIER:
            raise KamdError("invalid rspecifier " + rspecifier)
        if kind not in _AT and kind not in _ARK:
            raise KamdError("unknown table object type " + kind)
        self.kind, self.rspecifier = kind, rspecifier

    def __iter__(self):
        last = None
        for key, val in (self._background() if self.opts["background"] else self._entries()):
            if self.opts["sorted"] and last is not None and key <= last:
                raise KamdError("rspecifier %s: keys are not in sorted order (%s after %s)" % (self.rspecifier, key, last))
            last = key
            yield key, val

    def _background(self):
        """The `bg` option (SequentialTableReaderBackgroundImpl, util/kaldi-table-inl.h:1080-1290): the next object is
        read by a second thread while the consumer works on the current one -- one object ahead, like the reference;
        what the reading thread raises is raised where the object would have been delivered."""
        import queue
        import threading
        q = queue.Queue(maxsize=1)
        stop = threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                for kv in self._entries():
                    if not put(("item", kv)):
                        return
                put(("end", None))
            except BaseException as e:           # delivered to the consumer
                put(("error", e))

        t = threading.Thread(target=produce, daemon=True)
        t.start()
        try:
            while True:
                what, v = q.get()
                if what == "item":
                    yield v
                elif what == "error":
                    raise v
                else:
                    return
        finally:
            stop.set()                            # a consumer that stops early releases the reader
            t.join()

    def _entries(self):
        if self.type == ARCHIVE:
            if self.kind not in _ARK:
                raise KamdError("archives of %s objects are not supported (use an scp)" % self.kind)
            with Input(self.rx) as (path, off):
                if off:
                    raise KamdError("an archive rspecifier cannot carry an offset: " + self.rx)
                yield from _ARK[self.kind](path)
            return
        if self.kind not in _AT:
            raise KamdError("script files of %s objects are not supported (use an archive)" % self.kind)
        for key, rx in read_script_file(self.rx):
            try:
                with Input(rx) as (path, off):
                    val = _AT[self.kind](path, off)
            except (KamdError, OSError):
                if self.opts["permissive"]:
                    continue
                raise
            yield key, val


class RandomAccessTableReader:
    """HasKey / Value over an rspecifier (ivectors per utterance, utt2spk-style maps).  Script
    files are read on demand; archives are loaded on first use."""

    def __init__(self, rspecifier, kind):
        self.seq = SequentialTableReader(rspecifier, kind)
        self.kind = kind
        self._scp = dict(read_script_file(self.seq.rx)) if self.seq.type == SCRIPT else None
        self._all = None

    def _load(self):
        if self._all is None:
            self._all = dict(self.seq._entries())
        return self._all

    def has_key(self, key):
        return key in (self._scp if self._scp is not None else self._load())

    __contains__ = has_key

    def value(self, key):
        if self._scp is None:
            return self._load()[key]
        with Input(self._scp[key]) as (path, off):
            return _AT[self.kind](path, off)

    __getitem__ = value


def _int32_bytes(key, v, binary):
    v = np.asarray(v, np.int32)
    if not binary:
        return (key + " " + "".join("%d " % x for x in v) + "\n").encode()
    body = np.zeros(v.size, np.dtype([("s", "u1"), ("v", "<i4")]))       # BasicVectorHolder::Write
    body["s"], body["v"] = 4, v
    return key.encode() + b" \0B\x04" + np.int32(v.size).tobytes() + body.tobytes()


class TableWriter:
    """TableWriter("ark:lat.1" | "ark,t:-" | "ark:| gzip -c > lat.1.gz" | "ark,scp:f.ark,f.scp" | "scp:f.scp", kind)
    kind: "matrix" | "dmatrix" | "int32" | "lattice" | "compact_lattice".  A script-only wspecifier names one
    output file per key, as TableWriterScriptImpl does (no key inside the file)."""

    def __init__(self, wspecifier, kind, acoustic_scale=1.0):
        self.type, self.ark, self.scp, self.opts = classify_wspecifier(wspecifier)
        if self.type == NO_SPECIFIER:
            raise KamdError("invalid wspecifier " + wspecifier)
        if kind not in ("matrix", "dmatrix", "int32", "lattice", "compact_lattice", "raw"):
            raise KamdError("unknown table object type " + kind)
        self.kind, self.acoustic_scale, self.closed = kind, acoustic_scale, False
        self._scp_lines, self._tmp = [], None
        if self.type in (ARCHIVE, BOTH):
            wx = classify_wxfilename(self.ark)
            if wx == WX_NONE:
                raise KamdError("Invalid output filename format " + self.ark)
            if self.type == BOTH and wx != WX_FILE:
                raise KamdError("ark,scp output needs a real archive file, got " + self.ark)
            if wx == WX_FILE:
                self._path = self.ark
            else:                                   # pipe / stdout: spool, deliver at close()
                import tempfile
                fd, self._tmp = tempfile.mkstemp(prefix="kamd_wx_")
                os.close(fd)
                self._path = self._tmp
            open(self._path, "wb").close()
        else:
            self._targets = dict(read_script_file(self.scp))

    def _append(self, path, key, value, with_key=True):
        b = self.opts["binary"]
        if self.kind == "matrix":
            if not with_key:
                raise KamdError("script-only output of matrices is not supported")
            kio.write_matrix_ark(path, key, value, binary=b, append=True)
        elif self.kind == "int32":
            with open(path, "ab") as f:
                f.write(_int32_bytes(key, value, b))
        elif self.kind == "dmatrix":
            with open(path, "ab") as f:
                f.write((key.encode() + b" " if with_key else b"") + _dmatrix_bytes(value, b))
        elif self.kind == "raw":                     # value = the serialised object (after "key "), bytes
            with open(path, "ab") as f:
                f.write(key.encode() + b" " + value)
        elif self.kind == "lattice":
            kio.write_lattice(path, key, value, binary=b, append=True, acoustic_scale=self.acoustic_scale)
        else:
            value.write(path, key, binary=b, append=True, acoustic_scale=self.acoustic_scale)

    def write(self, key, value):
        if not key or any(c.isspace() for c in key):
            raise KamdError("TableWriter: invalid key \"%s\"" % key)       # IsToken
        if self.type == SCRIPT:
            if key not in self._targets:
                raise KamdError("TableWriter: key %s not in script file %s" % (key, self.scp))
            open(self._targets[key], "wb").close()
            self._append(self._targets[key], key, value)
            return
        start = os.path.getsize(self._path)
        self._append(self._path, key, value)
        if self.type == BOTH:
            self._scp_lines.append("%s %s:%d\n" % (key, self.ark, start + len(key.encode()) + 1))

    def close(self):
        if self.closed:
            return
        self.closed = True
        if self.type == BOTH:
            with open(self.scp, "w") as f:
                f.writelines(self._scp_lines)
        if self._tmp is not None:
            try:
                if classify_wxfilename(self.ark) == WX_STDOUT:
                    sys.stdout.flush()
                    with open(self._tmp, "rb") as f:
                        sys.stdout.buffer.write(f.read())
                    sys.stdout.flush()
                else:
                    with open(self._tmp, "rb") as f:
                        rc = subprocess.run(self.ark[1:], shell=True, stdin=f).returncode
                    if rc != 0:
                        raise KamdError("Pipe %s had nonzero return status %d" % (self.ark[1:], rc))
            finally:
                os.unlink(self._tmp)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


# ---- ParseOptions --------------------------------------------------------------------------
class ParseOptions:
    """po = ParseOptions(usage); po.register("beam", float, 16.0, "doc"); args = po.read(argv)
    Values: po["beam"] (names are normalised: lower case, '_' -> '-').  ParseOptions(prefix, parent)
    registers into the parent as --prefix.name (parse-options.cc:30-60)."""
    _TYPES = (bool, int, "uint", float, str)

    def __init__(self, usage_or_prefix="", parent=None):
        self.parent, self.prefix = parent, ""
        if parent is not None:
            self.prefix = usage_or_prefix + "."
        else:
            self.usage = usage_or_prefix
            self._types, self._values, self._docs = {}, {}, {}
            self.register("config", str, "", "Configuration file to read (this option may be repeated)")
            self.register("print-args", bool, True, "Print the command line arguments (to stderr)")
            self.register("help", bool, False, "Print out usage message")
            self.register("verbose", int, 0, "Verbose level (higher->more logging)")
        self.args = []

    @staticmethod
    def normalize(name):
        return name.replace("_", "-").lower()

    def register(self, name, typ, default, doc=""):
        if self.parent is not None:
            return self.parent.register(self.prefix + name, typ, default, doc)
        if typ not in self._TYPES:
            raise KamdError("ParseOptions: unsupported option type %r" % (typ,))
        key = self.normalize(name)
        self._types[key], self._values[key], self._docs[key] = typ, default, (name, doc)

    def __getitem__(self, name):
        if self.parent is not None:
            return self.parent[self.prefix + name]
        return self._values[self.normalize(name)]

    def was_given(self, name):
        """True iff the option appeared on the command line or in a --config file (not in the reference: used to
        warn where this implementation's default differs from Kaldi's)."""
        if self.parent is not None:
            return self.parent.was_given(self.prefix + name)
        return self.normalize(name) in getattr(self, "_given", ())

    def _set(self, key, value, has_eq, where):
        if key not in self._types:
            raise KamdError("Invalid option %s" % where)
        if not hasattr(self, "_given"):
            self._given = set()
        self._given.add(key)
        t = self._types[key]
        if t is bool:
            if has_eq and value == "":
                raise KamdError("Invalid option --%s=" % key)
            v = value.lower()
            if v in ("true", "t", "1", ""):
                self._values[key] = True
            elif v in ("false", "f", "0"):
                self._values[key] = False
            else:
                raise KamdError("Invalid format for boolean argument [expected true or false]: " + value)
        elif t is str:
            if not has_eq:
                raise KamdError("Invalid option --%s (option format is --x=y)." % key)
            self._values[key] = value
        elif t is float:
            try:
                self._values[key] = float(value)
            except ValueError:
                raise KamdError("Invalid floating-point option \"%s\"" % value)
        else:
            try:
                x = int(value, 10)
            except ValueError:
                raise KamdError("Invalid integer option \"%s\"" % value)
            lo, hi = (0, 2**32 - 1) if t == "uint" else (-2**31, 2**31 - 1)
            if not lo <= x <= hi or value != value.strip():
                raise KamdError("Invalid integer option \"%s\"" % value)
            self._values[key] = x

    @staticmethod
    def _split(arg):
        eq = arg.find("=")
        if eq < 0:
            return arg[2:], "", False
        if eq == 2:
            raise KamdError("Invalid option (no key): " + arg)
        return arg[2:eq], arg[eq + 1:], True

    def read_config_file(self, filename):
        try:
            lines = open(filename).read().split("\n")
        except OSError:
            raise KamdError("Cannot open config file: " + filename)
        for n, line in enumerate(lines):
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            if not line.startswith("--"):
                raise KamdError("Reading config file %s: line %d does not look like a line from a Kaldi command-line "
                                "program's config file: should be of the form --x=y." % (filename, n + 1))
            k, v, eq = self._split(line)
            self._set(self.normalize(k), v.strip(), eq, "%s in config file %s" % (line, filename))

    def read(self, argv):
        """argv[0] is the program name.  Returns the positional arguments (also self.args)."""
        for a in argv[1:]:                                  # first pass: config files
            if a.startswith("--"):
                if a == "--":
                    break
                k, v, _ = self._split(a)
                if self.normalize(k) == "config":
                    self.read_config_file(v.strip())
        i, dd = 1, False
        while i < len(argv):
            a = argv[i]
            if not a.startswith("--"):
                break
            i += 1
            if a == "--":
                dd = True
                break
            k, v, eq = self._split(a)
            self._set(self.normalize(k), v.strip(), eq, a)
        self.args = []
        for a in argv[i:]:
            if a == "--" and not dd:
                dd = True
            else:
                self.args.append(a)
        if self._values["help"]:
            self.print_usage()
            sys.exit(0)
        return self.args

    def num_args(self):
        return len(self.args)

    def get_arg(self, i):
        """1-based, like ParseOptions::GetArg"""
        if not 1 <= i <= len(self.args):
            raise KamdError("ParseOptions::GetArg, invalid index %d" % i)
        return self.args[i - 1]

    def print_usage(self, file=sys.stderr):
        print("\n" + self.usage + "\nOptions:", file=file)
        for k in sorted(self._docs):
            name, doc = self._docs[k]
            print("  --%-25s : %s (%s, default = %r)" % (name, doc, getattr(self._types[k], "__name__", self._types[k]), self._values[k]), file=file)
