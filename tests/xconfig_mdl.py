"""Test infrastructure: a nnet3 config text (what steps/nnet3/xconfig_to_configs.py writes as final.config;
the fixtures tests/golden/nnet/*.final.config come from the reference's own generator, see
tools/gen_xconfig_golden.py) -> (a) a binary final.mdl with random parameters whose config section is
exactly the node lines of that text, written component by component like the reference's Write functions
(the primitives of tests/mdl_writer.py), and (b) a DIRECT float64 evaluation of the same graph, node by
node and descriptor by descriptor, that shares no code with kaldi_amd/mdl.py's fused-layer compile:

  descriptors   nnet3/nnet-descriptor.h: Offset(x, k)[t] = x[t + k]; Append = column concatenation;
                Sum; Scale(s, x); ReplaceIndex(ivector, t, 0) = the utterance's i-vector at every t
  components    nnet-simple-component.cc: Affine / NaturalGradientAffine / FixedAffine y = W x + b (:1234,
                :3376), Linear y = W x (:3209), RectifiedLinear (:957), NoOp (:436);
                nnet-tdnn-component.cc:181-212 y[t] = sum_i W_i x[t + o_i] + b;
                nnet-normalize-component.cc:226-245,453-464 BatchNorm test mode y = x * scale + offset with
                scale = (var + eps)^-0.5 * target_rms, offset = -mean * scale;
                nnet-general-component.cc GeneralDropout test mode = identity
  input         frames outside [0, T) repeat the first / last frame (nnet-am-decodable-simple.cc:147-160)
"""
import re

import numpy as np

from tests.mdl_writer import boolean, f32, f64, i32, int_vector, mat, tok, transition_model, updatable_common, vec


def _parse_descriptor(text):
    """Descriptor text -> ('node', name) | ('Offset', d, k) | ('Append', [d..]) | ('Sum', a, b) | ('Scale', s, d) |
    ('ReplaceIndex', name): this file's own parser (the evaluator shares nothing with kaldi_amd/mdl.py)."""
    toks = re.findall(r"[A-Za-z_][\w.\-]*|-?\d+(?:\.\d*)?(?:[eE][-+]?\d+)?|[(),]", text)
    pos = 0

    def parse():
        nonlocal pos
        head = toks[pos]
        pos += 1
        if pos >= len(toks) or toks[pos] != "(":
            return ("node", head)
        pos += 1
        args = []
        while toks[pos] != ")":
            if toks[pos] == ",":
                pos += 1
            else:
                args.append(parse())
        pos += 1
        if head == "Offset":
            return ("Offset", args[0], int(args[1][1]))
        if head == "Append":
            return ("Append", args)
        if head == "Sum":
            assert len(args) == 2
            return ("Sum", args[0], args[1])
        if head == "Scale":
            return ("Scale", float(args[0][1]), args[1])
        if head == "ReplaceIndex":
            assert args[1][1] == "t" and int(args[2][1]) == 0
            return ("ReplaceIndex", args[0][1])
        raise ValueError("descriptor " + head)

    d = parse()
    assert pos == len(toks), text
    return d


def parse_config(text):
    """-> (inputs {name: dim}, components {name: (type, {key: value})}, nodes [(name, component, descriptor text)],
    outputs {name: descriptor text}, node_lines [the input- / component- / output-node lines verbatim])."""
    inputs, comps, nodes, outputs, node_lines = {}, {}, [], {}, []
    for line in text.split("\n"):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        kind, rest = line.split(" ", 1)
        if kind == "component":
            kv = dict(re.findall(r"([\w\-]+)=(\S+)", rest))
            comps[kv["name"]] = (kv["type"], kv)
            continue
        node_lines.append(line)
        if kind == "input-node":
            kv = dict(re.findall(r"([\w\-]+)=(\S+)", rest))
            inputs[kv["name"]] = int(kv["dim"])
        elif kind == "component-node":
            m = re.match(r"name=(\S+) component=(\S+) input=(.*)$", rest)
            nodes.append((m.group(1), m.group(2), m.group(3)))
        elif kind == "output-node":
            m = re.match(r"name=(\S+) input=(.*?)( objective=\S+)?$", rest)
            outputs[m.group(1)] = m.group(2)
        else:
            raise ValueError("unsupported config line: " + line)
    return inputs, comps, nodes, outputs, node_lines


def _desc_dim(d, dims):
    if d[0] == "node":
        return dims[d[1]]
    if d[0] == "Offset":
        return _desc_dim(d[1], dims)
    if d[0] == "Append":
        return sum(_desc_dim(p, dims) for p in d[1])
    if d[0] == "Sum":
        return _desc_dim(d[1], dims)
    if d[0] == "Scale":
        return _desc_dim(d[2], dims)
    if d[0] == "ReplaceIndex":
        return dims[d[1]]
    raise ValueError(d)


def random_params(text, seed=0):
    """Random parameters for every component of the config: {component name: dict}."""
    inputs, comps, nodes, _, _ = parse_config(text)
    rng = np.random.default_rng(seed)
    dims = dict(inputs)
    params = {}
    for name, cname, desc in nodes:
        typ, kv = comps[cname]
        in_dim = _desc_dim(_parse_descriptor(desc), dims)
        p = {"type": typ}
        if typ == "FixedAffineComponent":            # matrix=lda.mat: [dim x (dim + 1)], square LDA-like transform
            out_dim = in_dim
            p["W"] = (rng.standard_normal((out_dim, in_dim)) / np.sqrt(in_dim)).astype(np.float32)
            p["b"] = (0.1 * rng.standard_normal(out_dim)).astype(np.float32)
        elif typ in ("NaturalGradientAffineComponent", "AffineComponent"):
            assert int(kv["input-dim"]) == in_dim, (name, in_dim)
            out_dim = int(kv["output-dim"])
            p["W"] = (rng.standard_normal((out_dim, in_dim)) / np.sqrt(in_dim)).astype(np.float32)
            p["b"] = (0.1 * rng.standard_normal(out_dim)).astype(np.float32)
        elif typ == "LinearComponent":
            assert int(kv["input-dim"]) == in_dim, (name, in_dim)
            out_dim = int(kv["output-dim"])
            p["W"] = (rng.standard_normal((out_dim, in_dim)) / np.sqrt(in_dim)).astype(np.float32)
        elif typ == "TdnnComponent":
            assert int(kv["input-dim"]) == in_dim, (name, in_dim)
            out_dim = int(kv["output-dim"])
            offs = [int(x) for x in kv["time-offsets"].split(",")]
            p["offsets"] = offs
            k = in_dim * len(offs)
            p["W"] = (rng.standard_normal((out_dim, k)) / np.sqrt(k)).astype(np.float32)
            p["b"] = None if kv.get("use-bias", "true") == "false" else (0.1 * rng.standard_normal(out_dim)).astype(np.float32)
        elif typ == "BatchNormComponent":
            out_dim = int(kv["dim"])
            assert out_dim == in_dim
            p["mean"] = (0.2 * rng.standard_normal(out_dim)).astype(np.float32)
            p["var"] = rng.uniform(0.4, 2.0, out_dim).astype(np.float32)
            p["eps"], p["target_rms"] = 1e-3, float(kv.get("target-rms", 1.0))
        elif typ in ("RectifiedLinearComponent", "GeneralDropoutComponent", "NoOpComponent", "LogSoftmaxComponent"):
            out_dim = int(kv["dim"])
            assert out_dim == in_dim, (name, out_dim, in_dim)
        else:
            raise ValueError("component type %s not handled by this writer" % typ)
        p["dim"] = out_dim
        dims[name] = out_dim
        params[cname] = p
    return params


def _component_bytes(p):
    typ = p["type"]
    if typ == "FixedAffineComponent":               # nnet-simple-component.cc:3406-3413
        return tok("<FixedAffineComponent>") + tok("<LinearParams>") + mat(p["W"]) + tok("<BiasParams>") + vec(p["b"]) + \
            tok("</FixedAffineComponent>")
    if typ in ("NaturalGradientAffineComponent", "AffineComponent"):      # :2933-2955
        b = updatable_common(typ) + tok("<LinearParams>") + mat(p["W"]) + tok("<BiasParams>") + vec(p["b"])
        if typ == "NaturalGradientAffineComponent":
            b += tok("<RankIn>") + i32(20) + tok("<RankOut>") + i32(80) + tok("<UpdatePeriod>") + i32(4)
            b += tok("<NumSamplesHistory>") + f32(2000.0) + tok("<Alpha>") + f32(4.0)
        return b + tok("</%s>" % typ)
    if typ == "LinearComponent":                    # :3159-3186
        b = updatable_common(typ) + tok("<Params>") + mat(p["W"]) + tok("<OrthonormalConstraint>") + f32(-1.0)
        b += tok("<UseNaturalGradient>") + boolean(True) + tok("<RankInOut>") + i32(20) + i32(80)
        b += tok("<Alpha>") + f32(4.0) + tok("<NumSamplesHistory>") + f32(2000.0) + tok("<UpdatePeriod>") + i32(4)
        return b + tok("</LinearComponent>")
    if typ == "TdnnComponent":                      # nnet-tdnn-component.cc:379-405
        b = updatable_common(typ) + tok("<TimeOffsets>") + int_vector(p["offsets"]) + tok("<LinearParams>") + mat(p["W"])
        b += tok("<BiasParams>") + vec(p["b"] if p["b"] is not None else np.zeros(0))
        b += tok("<OrthonormalConstraint>") + f32(-1.0) + tok("<UseNaturalGradient>") + boolean(True)
        b += tok("<NumSamplesHistory>") + f32(2000.0) + tok("<AlphaInOut>") + f32(4.0) + f32(4.0)
        return b + tok("<RankInOut>") + i32(20) + i32(80) + tok("</TdnnComponent>")
    if typ in ("RectifiedLinearComponent", "LogSoftmaxComponent"):     # NonlinearComponent::Write, nnet-component-itf.cc:542-600
        z = np.zeros(p["dim"], np.float32)
        b = tok("<%s>" % typ) + tok("<Dim>") + i32(p["dim"]) + tok("<ValueAvg>") + vec(z)
        b += tok("<DerivAvg>") + vec(z) + tok("<Count>") + f64(0.0) + tok("<OderivRms>") + vec(z)
        b += tok("<OderivCount>") + f64(0.0) + tok("<NumDimsSelfRepaired>") + f64(0.0)
        b += tok("<NumDimsProcessed>") + f64(0.0) + tok("<SelfRepairScale>") + f32(1e-5)
        return b + tok("</%s>" % typ)
    if typ == "BatchNormComponent":                 # nnet-normalize-component.cc:614-640
        b = tok("<BatchNormComponent>") + tok("<Dim>") + i32(p["dim"]) + tok("<BlockDim>") + i32(p["dim"])
        b += tok("<Epsilon>") + f32(p["eps"]) + tok("<TargetRms>") + f32(p["target_rms"]) + tok("<TestMode>") + boolean(False)
        b += tok("<Count>") + f64(1000.0) + tok("<StatsMean>") + vec(p["mean"]) + tok("<StatsVar>") + vec(p["var"])
        return b + tok("</BatchNormComponent>")
    if typ == "GeneralDropoutComponent":            # nnet-general-component.cc:1641-1656
        b = tok("<GeneralDropoutComponent>") + tok("<Dim>") + i32(p["dim"]) + tok("<BlockDim>") + i32(p["dim"])
        return b + tok("<TimePeriod>") + i32(0) + tok("<DropoutProportion>") + f32(0.0) + tok("</GeneralDropoutComponent>")
    if typ == "NoOpComponent":                      # nnet-simple-component.cc:475-482
        return tok("<NoOpComponent>") + tok("<Dim>") + i32(p["dim"]) + tok("<BackpropScale>") + f32(1.0) + tok("</NoOpComponent>")
    raise ValueError(typ)


def write_mdl_from_config(path, text, params, priors, num_units):
    """Binary final.mdl: TransitionModel, then Nnet::Write (nnet3/nnet-nnet.cc:630-657: the node lines, an empty line, the
    components in the order of the config), then AmNnetSimple's context and priors (am-nnet-simple.cc:56-66)."""
    _, comps, _, _, node_lines = parse_config(text)
    tm, id2pdf, tid_phone = transition_model(num_units)
    out = b"\0B" + tm + tok("<Nnet3>") + b"\n" + ("\n".join(node_lines) + "\n\n").encode()
    names = [c for c in comps if c in params]
    out += tok("<NumComponents>") + i32(len(names))
    for name in names:
        out += tok("<ComponentName>") + tok(name) + _component_bytes(params[name])
    out += tok("</Nnet3>") + tok("<LeftContext>") + i32(0) + tok("<RightContext>") + i32(0)
    out += tok("<Priors>") + vec(priors)
    open(path, "wb").write(out)
    return id2pdf, tid_phone


def evaluate(text, params, feats, ivector, priors=None, acoustic_scale=1.0, subsampling=3, output="output", margin=64):
    """float64 evaluation of output node `output` at t = 0, subsampling, 2 subsampling, ... < T.  Every node is
    evaluated on the time range [-margin, T + margin); values that would need frames beyond that range are NaN and
    must not reach the output (margin >= the model's context)."""
    inputs, comps, nodes, outputs, _ = parse_config(text)
    T = feats.shape[0]
    lo, n = -margin, T + 2 * margin
    tt = np.clip(np.arange(lo, lo + n), 0, T - 1)
    val = {"input": feats.astype(np.float64)[tt]}
    if "ivector" in inputs:
        val["ivector"] = np.tile(np.asarray(ivector, np.float64)[None, :], (n, 1))
    node_of = {name: (cname, desc) for name, cname, desc in nodes}

    def shift(x, k):                                   # y[t] = x[t + k]
        y = np.full_like(x, np.nan)
        if k >= 0:
            y[:n - k] = x[k:]
        else:
            y[-k:] = x[:n + k]
        return y

    def desc_val(d):
        if d[0] == "node":
            return node_val(d[1])
        if d[0] == "Offset":
            return shift(desc_val(d[1]), d[2])
        if d[0] == "Append":
            return np.concatenate([desc_val(p) for p in d[1]], axis=1)
        if d[0] == "Sum":
            return desc_val(d[1]) + desc_val(d[2])
        if d[0] == "Scale":
            return d[1] * desc_val(d[2])
        if d[0] == "ReplaceIndex":
            return val[d[1]]
        raise ValueError(d)

    def node_val(name):
        if name in val:
            return val[name]
        cname, desc = node_of[name]
        p = params[cname]
        x = desc_val(_parse_descriptor(desc))
        typ = p["type"]
        if typ in ("FixedAffineComponent", "NaturalGradientAffineComponent", "AffineComponent"):
            y = x @ p["W"].astype(np.float64).T + p["b"].astype(np.float64)
        elif typ == "LinearComponent":
            y = x @ p["W"].astype(np.float64).T
        elif typ == "TdnnComponent":
            d = x.shape[1]
            y = np.zeros((n, p["dim"]))
            for i, o in enumerate(p["offsets"]):
                y += shift(x, o) @ p["W"][:, i * d:(i + 1) * d].astype(np.float64).T
            if p["b"] is not None:
                y += p["b"].astype(np.float64)
        elif typ == "RectifiedLinearComponent":
            y = np.where(np.isnan(x), np.nan, np.maximum(x, 0.0))
        elif typ == "BatchNormComponent":
            scale = (np.maximum(p["var"].astype(np.float64), 0.0) + p["eps"]) ** -0.5 * p["target_rms"]
            y = x * scale - p["mean"].astype(np.float64) * scale
        elif typ in ("GeneralDropoutComponent", "NoOpComponent"):
            y = x
        elif typ == "LogSoftmaxComponent":
            mx = np.max(x, axis=1, keepdims=True)
            y = x - mx - np.log(np.sum(np.exp(x - mx), axis=1, keepdims=True))
        else:
            raise ValueError(typ)
        val[name] = y
        return y

    out = desc_val(_parse_descriptor(outputs[output]))
    rows = np.arange(0, T, subsampling) - lo
    y = out[rows]
    assert not np.isnan(y).any(), "margin smaller than the model's context"
    if priors is not None and len(priors):
        y = y - np.log(np.asarray(priors, np.float64))
    return y * acoustic_scale
