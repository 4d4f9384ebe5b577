"""Reader for binary Kaldi `final.mdl` files of chain TDNN / TDNN-F models: TransitionModel
(hmm/transition-model.cc:394-420, hmm/hmm-topology.cc Read) + AmNnetSimple (nnet3/am-nnet-simple.cc:
44-54) = Nnet (nnet3/nnet-nnet.cc:586-628: text config lines, then the components) + priors.

What it returns is what the device path consumes: a `kaldi_amd.nnet.Model` (fused layers =
what CollapseModel (nnet3/nnet-utils.cc:2006) + the decodable's "-log prior, * acoustic scale"
amount to at test time), `id2pdf` (TransitionModel::id2pdf_id_, index 0 unused) and `tid_phone`
(the table DeterminizeLatticePhonePruned reads from the transition model).

The component bodies are parsed with a generic tokenizer of Kaldi's binary stream (tokens,
size-prefixed basic types, FM / FV objects; base/io-funcs-inl.h, matrix/kaldi-matrix.cc:1378-1404),
so fields this reader does not use (learning rates, natural-gradient settings, statistics)
are skipped by structure, not by a per-version field list.  The network graph is matched
against the xconfig patterns of the recipes (steps/libs/nnet3/xconfig: relu-batchnorm-dropout-
layer, tdnnf-layer, linear-component, prefinal-layer, output-layer); anything else raises.
PARITY UNPINNED: no model file exists in the reference tree; the layouts are restated from
the in-tree Read/Write code and exercised by a writer that follows the Write functions
(tests/mdl_writer.py, tests/test_mdl.py)."""
import math
import re
import struct

import numpy as np

from .nnet import Layer, Model


class MdlError(Exception):
    pass


class _Stream:
    def __init__(self, data):
        self.b, self.p = data, 0

    def peek(self, n=1):
        return self.b[self.p:self.p + n]

    def take(self, n):
        if self.p + n > len(self.b):
            raise MdlError("unexpected end of file")
        r = self.b[self.p:self.p + n]
        self.p += n
        return r

    def token(self):
        e = self.b.index(b" ", self.p)
        t = self.b[self.p:e].decode()
        self.p = e + 1
        return t

    def expect(self, tok):
        t = self.token()
        if t != tok:
            raise MdlError("expected %s, got %s" % (tok, t))

    def i32(self):
        if self.take(1) != b"\x04":
            raise MdlError("int32 expected at byte %d" % self.p)
        return struct.unpack("<i", self.take(4))[0]

    def f32(self):
        if self.take(1) != b"\x04":
            raise MdlError("float expected at byte %d" % self.p)
        return struct.unpack("<f", self.take(4))[0]

    def int_vector(self):                      # WriteIntegerVector (base/io-funcs-inl.h:198-229)
        if self.take(1) != b"\x04":
            raise MdlError("integer vector expected")
        n = struct.unpack("<i", self.take(4))[0]
        return np.frombuffer(self.take(4 * n), "<i4").copy()

    def vector(self):                          # Vector<float>::Write: "FV " size data
        t = self.token()
        if t not in ("FV", "DV"):
            raise MdlError("vector expected, got %s" % t)
        n = self.i32()
        if t == "FV":
            return np.frombuffer(self.take(4 * n), "<f4").copy()
        return np.frombuffer(self.take(8 * n), "<f8").astype(np.float32)

    def matrix(self):
        t = self.token()
        if t not in ("FM", "DM"):
            raise MdlError("matrix expected, got %s" % t)
        r, c = self.i32(), self.i32()
        if t == "FM":
            return np.frombuffer(self.take(4 * r * c), "<f4").reshape(r, c).copy()
        return np.frombuffer(self.take(8 * r * c), "<f8").reshape(r, c).astype(np.float32)

    def generic_fields(self, end_token):
        """{token: [values]} until `end_token`: values are raw basic types, bools, vectors, matrices."""
        out, cur = {}, None
        while True:
            c = self.peek()
            if c == b"<":
                t = self.token()
                if t == end_token:
                    return out
                cur = out.setdefault(t, [])
                if t == "<TimeOffsets>":
                    cur.append(self.int_vector())
            elif c in (b"F", b"D") and self.peek(3)[1:2] in (b"M", b"V") and self.peek(3)[2:3] == b" ":
                cur.append(self.matrix() if self.peek(2)[1:2] == b"M" else self.vector())
            elif c in (b"T", b"F"):
                cur.append(self.take(1) == b"T")
            elif c in (b"\x04", b"\x08", b"\x01", b"\x02"):
                n = self.take(1)[0]
                cur.append(self.take(n))
            else:
                raise MdlError("cannot parse component body at byte %d (%r)" % (self.p, self.peek(8)))


def _f(raw):
    return struct.unpack("<f", raw)[0] if len(raw) == 4 else struct.unpack("<d", raw)[0]


def _i(raw):
    return struct.unpack("<i", raw)[0]


def read_transition_model(s):
    """-> (id2pdf [num_tids + 1], tid_phone [num_tids + 1]: phone of a transition-id that enters a
    phone (hmm-state 0, not a self-loop), 0 elsewhere)."""
    s.expect("<TransitionModel>")
    s.expect("<Topology>")                                 # binary form (hmm-topology.cc:208-227)
    phones = s.int_vector()
    phone2idx = s.int_vector()
    n = s.i32()
    is_hmm = True
    if n == -1:
        is_hmm = False
        n = s.i32()
    entries = []
    for _ in range(n):
        states = []
        for _ in range(s.i32()):
            fwd = s.i32()
            slf = fwd if is_hmm else s.i32()
            trans = [(s.i32(), s.f32()) for _ in range(s.i32())]
            states.append((fwd, slf, trans))
        entries.append(states)
    s.expect("</Topology>")
    tok = s.token()
    if tok not in ("<Triples>", "<Tuples>"):
        raise MdlError("expected <Triples> or <Tuples>, got " + tok)
    tuples = []
    for _ in range(s.i32()):
        ph, hs, fp = s.i32(), s.i32(), s.i32()
        sp = fp if tok == "<Triples>" else s.i32()
        tuples.append((ph, hs, fp, sp))
    s.expect("</Triples>" if tok == "<Triples>" else "</Tuples>")
    s.expect("<LogProbs>")
    s.vector()
    s.expect("</LogProbs>")
    s.expect("</TransitionModel>")
    id2pdf, tid_phone, tid2phone = [-1], [0], [0]           # ComputeDerived (transition-model.cc:144-188)
    for ph, hs, fp, sp in tuples:
        trans = entries[phone2idx[ph]][hs][2]
        for dst, _ in trans:
            self_loop = dst == hs                           # IsSelfLoop (:319-327)
            id2pdf.append(sp if self_loop else fp)
            tid_phone.append(ph if (hs == 0 and not self_loop) else 0)
            tid2phone.append(ph)                            # TransitionIdToPhone
    read_transition_model.tid2phone = np.asarray(tid2phone, np.int32)
    return np.asarray(id2pdf, np.int32), np.asarray(tid_phone, np.int32), phones


def _parse_descriptor(text):
    """nnet3 descriptor text -> nested tuples: ('node', name) | ('Offset', d, n) | ('Append', [..]) |
    ('Sum', a, b) | ('Scale', s, d) | ('ReplaceIndex', name, 't', 0)."""
    toks = re.findall(r"[A-Za-z_][\w.\-]*|-?\d+\.?\d*(?:[eE][-+]?\d+)?|[(),]", text)
    pos = [0]

    def parse():
        t = toks[pos[0]]
        pos[0] += 1
        if pos[0] < len(toks) and toks[pos[0]] == "(":
            pos[0] += 1
            args = []
            while toks[pos[0]] != ")":
                if toks[pos[0]] == ",":
                    pos[0] += 1
                    continue
                args.append(parse())
            pos[0] += 1
            if t == "Offset":
                return ("Offset", args[0], int(args[1][1]))
            if t == "Append":
                return ("Append", args)
            if t == "Sum":
                return ("Sum", args[0], args[1])
            if t == "Scale":
                return ("Scale", float(args[0][1]), args[1])
            if t == "ReplaceIndex":
                return ("ReplaceIndex", args[0][1])
            raise MdlError("unsupported descriptor " + t)
        return ("node", t)

    return parse()


# components that are a per-element affine map y = x * scale + offset at test time
_PER_ELEMENT = ("FixedScaleComponent", "FixedBiasComponent", "PerElementScaleComponent", "NaturalGradientPerElementScaleComponent",
                "PerElementOffsetComponent", "ScaleAndOffsetComponent")


def _per_element_map(typ, f, dim, name):
    """(scale, offset) of the component, float32 [dim] (nnet-simple-component.cc: FixedScale :3669, FixedBias :3740,
    PerElementScale :2021, PerElementOffset :2191-2225 and ScaleAndOffset :2400-2440 -- the latter two repeat a shorter
    parameter vector block-wise over the dimension)."""
    one, zero = np.ones(dim, np.float32), np.zeros(dim, np.float32)

    def full(v):
        v = np.asarray(v, np.float32).reshape(-1)
        if v.size == 0 or dim % v.size:
            raise MdlError("%s: parameter vector of %d elements for dimension %d" % (name, v.size, dim))
        return np.tile(v, dim // v.size)
    if typ == "FixedScaleComponent":
        return full(f["<Scales>"][0]), zero
    if typ == "FixedBiasComponent":
        return one, full(f["<Bias>"][0])
    if typ in ("PerElementScaleComponent", "NaturalGradientPerElementScaleComponent"):
        return full(f["<Params>"][0]), zero
    if typ == "PerElementOffsetComponent":
        return one, full(f["<Offsets>"][0])
    return full(f["<Scales>"][0]), full(f["<Offsets>"][0])


def _apply_per_element(L, scale, offset, name):
    """y = layer(x) * scale + offset, folded into the fused layer: into the GEMM's rows while nothing but the affine map
    has been applied yet, into the post-ReLU per-element map (the BatchNorm slot) afterwards."""
    if L.bypass_layer != -2 or L.log_softmax:
        raise MdlError("a per-element map after a bypass / log-softmax is not representable (%s)" % name)
    if not L.relu and L.bn_scale is None:
        L.W = np.ascontiguousarray(L.W * scale[:, None], np.float32)
        b = np.zeros(L.out_dim, np.float32) if L.bias is None else L.bias
        L.bias = (b * scale + offset).astype(np.float32)
    elif L.bn_scale is None:
        L.bn_scale, L.bn_offset = scale.copy(), offset.copy()
    else:
        L.bn_offset = (L.bn_offset * scale + offset).astype(np.float32)
        L.bn_scale = (L.bn_scale * scale).astype(np.float32)


def read_mdl(path, acoustic_scale=1.0, frame_subsampling_factor=3):
    s = _Stream(open(path, "rb").read())
    if s.take(2) != b"\0B":
        raise MdlError("binary Kaldi file expected (text-mode models: nnet3-am-copy --binary=true)")
    id2pdf, tid_phone, _ = read_transition_model(s)
    tid2phone = read_transition_model.tid2phone             # every transition-id's phone (endpointing)
    s.expect("<Nnet3>")
    # config section: text lines until an empty line (nnet-nnet.cc:601-610)
    end = s.b.index(b"\n\n", s.p)
    config = s.b[s.p:end].decode().strip().split("\n")
    s.p = end + 2
    s.expect("<NumComponents>")
    comps = {}
    for _ in range(s.i32()):
        s.expect("<ComponentName>")
        name = s.token()
        typ = s.token()                                    # "<TdnnComponent>"
        comps[name] = (typ[1:-1], s.generic_fields("</" + typ[1:]))
    s.expect("</Nnet3>")
    s.expect("<LeftContext>"); s.i32()
    s.expect("<RightContext>"); s.i32()
    s.expect("<Priors>")
    priors = s.vector()
    # ---- graph: component-node name=X component=C input=DESCRIPTOR
    nodes, inputs, out_node = {}, {}, None
    for line in config:
        def field(key):
            m = re.search(r"\b%s=(\S+)" % key, line)
            return m.group(1) if m else None
        if line.startswith("input-node"):
            inputs[field("name")] = int(field("dim"))
        elif line.startswith("component-node"):
            nodes[field("name")] = (field("component"), _parse_descriptor(line.split("input=", 1)[1]))
        elif line.startswith("output-node") and field("name") == "output":
            out_node = _parse_descriptor(line.split("input=", 1)[1].split(" objective=")[0])
    if out_node is None or "input" not in inputs:
        raise MdlError("no output-node named 'output' / input-node named 'input'")
    layers, layer_of = [], {}            # node name -> index of the fused layer whose output it is

    def resolve(desc):
        """-> index of the fused layer this plain descriptor denotes (-1 = network input)."""
        if desc[0] != "node":
            raise MdlError("unsupported input descriptor %r" % (desc,))
        if desc[1] == "input":
            return -1
        return build(desc[1])

    def build(name):
        if name in layer_of:
            return layer_of[name]
        cname, desc = nodes[name]
        typ, f = comps[cname]
        if typ in ("FixedAffineComponent", "NaturalGradientAffineComponent", "AffineComponent", "LinearComponent", "TdnnComponent"):
            W = f["<Params>"][0] if typ == "LinearComponent" else f["<LinearParams>"][0]
            bias = None if typ == "LinearComponent" else f["<BiasParams>"][0]
            if bias is not None and bias.size == 0:
                bias = None
            ivector_dim = 0
            slice_layers = slice_dims = None                # Append over different producers
            col_scales = []                                 # one per appended slice, when a Scale(...) descriptor feeds it
            whole_scale = 1.0
            if desc[0] == "Scale":                          # input=Scale(s, x): y = W (s x) + b
                whole_scale, desc = desc[1], desc[2]
            if typ == "TdnnComponent":
                offsets = [int(x) for x in f["<TimeOffsets>"][0]]
                src = resolve(desc)
            elif desc[0] == "Append":                       # lda: Append(Offset(input,-1), input, Offset(input,1), ReplaceIndex(ivector,t,0))
                offsets, src, producers = [], None, []
                for part in desc[1]:
                    if part[0] == "ReplaceIndex":
                        ivector_dim = inputs[part[1]]
                        continue
                    part_scale = 1.0
                    if part[0] == "Scale":                  # Scale(s, Offset(x, t)): folded into that slice's columns below
                        part_scale, part = part[1], part[2]
                    off, inner = (part[2], part[1]) if part[0] == "Offset" else (0, part)
                    if inner[0] == "Scale":
                        part_scale, inner = part_scale * inner[1], inner[2]
                    col_scales.append(part_scale)
                    idx = resolve(inner)
                    producers.append(idx)
                    src = idx
                    offsets.append(off)
                if len(set(producers)) > 1:                 # Append over different producers: a multi-input layer
                    if ivector_dim:
                        raise MdlError("Append over different producers together with an ivector is not supported (%s)" % name)
                    slice_layers = producers
                    slice_dims = [inputs["input"] if q == -1 else layers[q].out_dim for q in producers]
                    src = -1
            elif desc[0] == "Offset":                       # input=Offset(x, t)
                offsets, src = [desc[2]], resolve(desc[1])
            else:
                offsets, src = [0], resolve(desc)
            in_dim = inputs["input"] if src == -1 else layers[src].out_dim
            widths = [in_dim] * len(offsets) if slice_layers is None else slice_dims
            if slice_layers is not None:
                in_dim = int(sum(slice_dims))
            if W.shape[1] != int(sum(widths)) + ivector_dim:
                raise MdlError("%s: parameter shape %s does not match its input" % (name, W.shape))
            if whole_scale != 1.0 or (col_scales and any(c != 1.0 for c in col_scales)):
                W = np.array(W, np.float32)
                cs = np.full(len(offsets), whole_scale, np.float32) if not col_scales else np.asarray(col_scales, np.float32) * np.float32(whole_scale)
                W[:, :int(sum(widths))] *= np.repeat(cs, widths)[None, :]
                if whole_scale != 1.0 and ivector_dim:      # Scale(s, Append(.., ReplaceIndex(ivector, t, 0))) scales the i-vector too
                    W[:, int(sum(widths)):] *= np.float32(whole_scale)
            layers.append(Layer(name, in_dim, W.shape[0], offsets, src, np.ascontiguousarray(W, np.float32),
                                None if bias is None else bias.astype(np.float32), ivector_dim=ivector_dim,
                                slice_layers=slice_layers, slice_dims=slice_dims))
            layer_of[name] = len(layers) - 1
            return layer_of[name]
        # components that act on the output of the layer they follow
        if typ == "NoOpComponent" and desc[0] == "Sum":      # tdnnf bypass: Sum(Scale(s, prev), this)
            a, b = desc[1], desc[2]
            scaled, plain = (a, b) if a[0] == "Scale" else (b, a)
            if scaled[0] != "Scale":                        # Sum(x, y): a residual connection with scale 1; the later layer is "this"
                ia, ib = resolve(a), resolve(b)
                plain, scaled = (a, ("Scale", 1.0, b)) if ia > ib else (b, ("Scale", 1.0, a))
            idx = resolve(plain)
            byp = resolve(scaled[2])
            L = layers[idx]
            if L.bypass_layer != -2:
                raise MdlError("two bypass connections on one layer (%s)" % name)
            L.bypass_layer, L.bypass_scale = byp, float(scaled[1])
            layer_of[name] = idx
            return idx
        idx = resolve(desc)
        L = layers[idx]
        if typ == "RectifiedLinearComponent":
            if L.relu or L.bn_scale is not None or L.bypass_layer != -2:
                raise MdlError("ReLU after batchnorm / bypass is not representable (%s)" % name)
            L.relu = True
        elif typ == "BatchNormComponent":
            if L.bypass_layer != -2:
                raise MdlError("batchnorm after a bypass is not representable (%s)" % name)
            mean, var = f["<StatsMean>"][0].astype(np.float32), f["<StatsVar>"][0].astype(np.float32)
            eps, rms = np.float32(_f(f["<Epsilon>"][0])), np.float32(_f(f["<TargetRms>"][0]))
            # ComputeDerived (nnet-normalize-component.cc:226-245)
            # (the power in double through libm, rounded to float: bit-identical with csrc/mdl.cc; the reference's own
            # powf differs from it by rounding in at most the rare double-rounding case)
            scale = np.array([math.pow(float(v), -0.5) for v in np.maximum(var, np.float32(0.0)) + eps], np.float32) * rms
            offset = -mean * scale
            if L.bn_scale is not None:                     # two per-element affine maps compose
                L.bn_offset = L.bn_offset * scale + offset
                L.bn_scale = L.bn_scale * scale
            else:
                L.bn_scale, L.bn_offset = scale.astype(np.float32), offset.astype(np.float32)
        elif typ in ("GeneralDropoutComponent", "DropoutComponent", "NoOpComponent", "ClipGradientComponent", "BackpropTruncationComponent"):
            pass                                           # identity at test time
        elif typ in _PER_ELEMENT:
            scale, offset = _per_element_map(typ, f, L.out_dim, name)
            _apply_per_element(L, scale, offset, name)
        elif typ == "LogSoftmaxComponent":
            L.log_softmax = True
        else:
            raise MdlError("unsupported component type %s (%s)" % (typ, name))
        layer_of[name] = idx
        return idx

    out_idx = resolve(out_node)
    if out_idx != len(layers) - 1:
        raise MdlError("the output node must be the last layer built")
    out = layers[out_idx]
    if priors.size:
        with np.errstate(divide="ignore"):
            out.post_offset = np.array([-math.log(float(v)) if v > 0 else math.inf for v in priors], np.float32)   # nnet-am-decodable-simple.cc:268-269
    out.post_scale = float(acoustic_scale)
    m = Model(layers, inputs["input"], inputs.get("ivector", 0), frame_subsampling_factor, out.out_dim,
              name=str(path))
    m.tid2phone = tid2phone
    return m, id2pdf, tid_phone


def read_mdl_native(path, acoustic_scale=1.0, frame_subsampling_factor=3):
    """The same three results through the library's own reader (`kamd_model_read`, csrc/mdl.cc: what a C / C++ host
    uses), copied out of the handle.  Bit-identical with read_mdl (tests/test_mdl.py, tests/test_xconfig_golden.py)."""
    import ctypes as C

    from . import _lib
    L = _lib.lib()
    h = L.kamd_model_read(str(path).encode(), float(acoustic_scale), int(frame_subsampling_factor))
    if not h:
        raise MdlError(L.kamd_last_error().decode())
    try:
        v = [C.c_int32() for _ in range(6)]
        L.kamd_model_info(h, *[C.byref(x) for x in v])
        n_layers, input_dim, ivector_dim, num_pdfs, num_tids, sub = [x.value for x in v]
        tabs = [np.zeros(num_tids + 1, np.int32) for _ in range(3)]
        L.kamd_model_transition_tables(h, *[t.ctypes.data_as(C.POINTER(C.c_int32)) for t in tabs])
        descs = L.kamd_model_layers(h)

        def arr(p, *shape):
            return None if not p else np.ctypeslib.as_array(p, shape=shape).copy()
        layers = []
        for i in range(n_layers):
            d = descs[i]
            cols = (d.in_dim if d.multi_input else d.n_offsets * d.in_dim) + d.ivector_dim
            layers.append(Layer("layer%d" % i, d.in_dim, d.out_dim, [int(d.offsets[j]) for j in range(d.n_offsets)], d.input_layer,
                                arr(d.W, d.out_dim, cols), arr(d.bias, d.out_dim), bool(d.relu), arr(d.bn_scale, d.out_dim),
                                arr(d.bn_offset, d.out_dim), d.bypass_layer, float(d.bypass_scale), d.ivector_dim,
                                arr(d.post_offset, d.out_dim), float(d.post_scale), bool(d.log_softmax),
                                [int(d.slice_layer[j]) for j in range(d.n_offsets)] if d.multi_input else None,
                                [int(d.slice_dim[j]) for j in range(d.n_offsets)] if d.multi_input else None))
    finally:
        L.kamd_model_destroy(h)
    m = Model(layers, input_dim, ivector_dim, sub, num_pdfs, name=str(path))
    m.tid2phone = tabs[2]
    return m, tabs[0], tabs[1]
