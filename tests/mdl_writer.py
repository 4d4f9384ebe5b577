"""Writes a kaldi_amd.nnet.Model as a binary Kaldi final.mdl, component by component, following
the reference's Write functions (test infrastructure for tests/test_mdl.py):
  TransitionModel::Write   hmm/transition-model.cc:422-453, HmmTopology::Write hmm-topology.cc:208-228
  Nnet::Write              nnet3/nnet-nnet.cc:630-657
  AmNnetSimple::Write      nnet3/am-nnet-simple.cc:56-66
  components               nnet-simple-component.cc:2933-2955 (NaturalGradientAffine), 3159-3186 (Linear),
                           3406-3413 (FixedAffine), 475-482 (NoOp); nnet-tdnn-component.cc:379-405;
                           nnet-component-itf.cc:302-326, 542-600 (Nonlinear); nnet-normalize-component.cc:614-640;
                           nnet-general-component.cc:1641-1656"""
import struct

import numpy as np


def tok(t):
    return t.encode() + b" "


def i32(x):
    return b"\x04" + struct.pack("<i", int(x))


def f32(x):
    return b"\x04" + struct.pack("<f", float(x))


def f64(x):
    return b"\x08" + struct.pack("<d", float(x))


def boolean(b):
    return b"T" if b else b"F"


def vec(v):
    v = np.ascontiguousarray(v, "<f4")
    return tok("FV") + i32(v.size) + v.tobytes()


def mat(m):
    m = np.ascontiguousarray(m, "<f4")
    return tok("FM") + i32(m.shape[0]) + i32(m.shape[1]) + m.tobytes()


def int_vector(v):
    v = np.ascontiguousarray(v, "<i4")
    return b"\x04" + struct.pack("<i", v.size) + v.tobytes()


def updatable_common(typ):
    return tok("<%s>" % typ) + tok("<MaxChange>") + f32(0.75) + tok("<LearningRate>") + f32(0.001)


def transition_model(num_units):
    """chain topology (one emitting state per phone, forward / self-loop pdf classes), phones 1..U"""
    phones = np.arange(1, num_units + 1)
    out = tok("<TransitionModel>") + tok("<Topology>")
    out += int_vector(phones) + int_vector(np.concatenate([[-1], np.zeros(num_units, np.int32)]))
    out += i32(-1) + i32(1)                    # extended format, one topology entry
    out += i32(2)                              # two states
    out += i32(0) + i32(1) + i32(2) + i32(0) + f32(0.5) + i32(1) + f32(0.5)     # state 0: pdf classes 0 / 1, self-loop first
    out += i32(-1) + i32(-1) + i32(0)          # state 1: final, no pdf, no transitions
    out += tok("</Topology>") + tok("<Tuples>") + i32(num_units)
    for p in phones:
        out += i32(p) + i32(0) + i32(2 * (p - 1)) + i32(2 * (p - 1) + 1)
    out += tok("</Tuples>") + tok("<LogProbs>") + vec(np.full(2 * num_units + 1, np.log(0.5), np.float32))
    out += tok("</LogProbs>") + tok("</TransitionModel>")
    id2pdf, tid_phone = [-1], [0]
    for p in phones:
        id2pdf += [2 * (p - 1) + 1, 2 * (p - 1)]     # tid 1: self-loop (pdf class 1), tid 2: forward
        tid_phone += [0, int(p)]
    return out, np.asarray(id2pdf, np.int32), np.asarray(tid_phone, np.int32)


def write_mdl(path, model, num_units):
    cfg = ["input-node name=input dim=%d" % model.input_dim]
    if model.ivector_dim:
        cfg.append("input-node name=ivector dim=%d" % model.ivector_dim)
    comps = []                                 # (name, bytes)
    final_name = {-1: "input"}
    for li, L in enumerate(model.layers):
        base = L.name or ("layer%d" % li)
        src = final_name[L.input_layer]
        if L.input_layer == -1 and (len(L.offsets) > 1 or L.ivector_dim):
            typ = "FixedAffineComponent"
            parts = [("Offset(%s, %d)" % (src, o) if o else src) for o in L.offsets]
            if L.ivector_dim:
                parts.append("ReplaceIndex(ivector, t, 0)")
            inp = "Append(%s)" % ", ".join(parts)
            body = tok("<FixedAffineComponent>") + tok("<LinearParams>") + mat(L.W) + tok("<BiasParams>") + \
                vec(L.bias if L.bias is not None else np.zeros(L.out_dim)) + tok("</FixedAffineComponent>")
        elif list(L.offsets) != [0]:
            typ, inp = "TdnnComponent", src
            body = updatable_common(typ) + tok("<TimeOffsets>") + int_vector(L.offsets) + tok("<LinearParams>") + mat(L.W)
            body += tok("<BiasParams>") + vec(L.bias if L.bias is not None else np.zeros(0))
            body += tok("<OrthonormalConstraint>") + f32(-1.0) + tok("<UseNaturalGradient>") + boolean(True)
            body += tok("<NumSamplesHistory>") + f32(2000.0) + tok("<AlphaInOut>") + f32(4.0) + f32(4.0)
            body += tok("<RankInOut>") + i32(20) + i32(80) + tok("</TdnnComponent>")
        elif L.bias is None:
            typ, inp = "LinearComponent", src
            body = updatable_common(typ) + tok("<Params>") + mat(L.W) + tok("<OrthonormalConstraint>") + f32(-1.0)
            body += tok("<UseNaturalGradient>") + boolean(True) + tok("<RankInOut>") + i32(20) + i32(80)
            body += tok("<Alpha>") + f32(4.0) + tok("<NumSamplesHistory>") + f32(2000.0) + tok("<UpdatePeriod>") + i32(4)
            body += tok("</LinearComponent>")
        else:
            typ, inp = "NaturalGradientAffineComponent", src
            body = updatable_common(typ) + tok("<LinearParams>") + mat(L.W) + tok("<BiasParams>") + vec(L.bias)
            body += tok("<RankIn>") + i32(20) + tok("<RankOut>") + i32(80) + tok("<UpdatePeriod>") + i32(4)
            body += tok("<NumSamplesHistory>") + f32(2000.0) + tok("<Alpha>") + f32(4.0)
            body += tok("</NaturalGradientAffineComponent>")
        last = li == len(model.layers) - 1
        name = "output.affine" if last else base + ".affine"
        comps.append((name, body))
        cfg.append("component-node name=%s component=%s input=%s" % (name, name, inp))
        cur = name
        if L.relu:
            n = base + ".relu"
            z = np.zeros(L.out_dim, np.float32)
            body = tok("<RectifiedLinearComponent>") + tok("<Dim>") + i32(L.out_dim) + tok("<ValueAvg>") + vec(z)
            body += tok("<DerivAvg>") + vec(z) + tok("<Count>") + f64(0.0) + tok("<OderivRms>") + vec(z)
            body += tok("<OderivCount>") + f64(0.0) + tok("<NumDimsSelfRepaired>") + f64(0.0)
            body += tok("<NumDimsProcessed>") + f64(0.0) + tok("<SelfRepairScale>") + f32(1e-5)
            body += tok("</RectifiedLinearComponent>")
            comps.append((n, body)); cfg.append("component-node name=%s component=%s input=%s" % (n, n, cur)); cur = n
        if L.bn_scale is not None:
            n = base + ".batchnorm"
            eps = 1e-3
            scale, offset = L.bn_scale.astype(np.float64), L.bn_offset.astype(np.float64)
            var = scale ** -2 - eps
            mean = -offset / scale
            body = tok("<BatchNormComponent>") + tok("<Dim>") + i32(L.out_dim) + tok("<BlockDim>") + i32(L.out_dim)
            body += tok("<Epsilon>") + f32(eps) + tok("<TargetRms>") + f32(1.0) + tok("<TestMode>") + boolean(False)
            body += tok("<Count>") + f64(1000.0) + tok("<StatsMean>") + vec(mean) + tok("<StatsVar>") + vec(var)
            body += tok("</BatchNormComponent>")
            comps.append((n, body)); cfg.append("component-node name=%s component=%s input=%s" % (n, n, cur)); cur = n
            n = base + ".dropout"
            body = tok("<GeneralDropoutComponent>") + tok("<Dim>") + i32(L.out_dim) + tok("<BlockDim>") + i32(L.out_dim)
            body += tok("<TimePeriod>") + i32(0) + tok("<DropoutProportion>") + f32(0.0) + tok("</GeneralDropoutComponent>")
            comps.append((n, body)); cfg.append("component-node name=%s component=%s input=%s" % (n, n, cur)); cur = n
        if L.bypass_layer != -2:
            n = base + ".noop"
            body = tok("<NoOpComponent>") + tok("<Dim>") + i32(L.out_dim) + tok("<BackpropScale>") + f32(1.0) + tok("</NoOpComponent>")
            comps.append((n, body))
            cfg.append("component-node name=%s component=%s input=Sum(Scale(%g, %s), %s)" % (
                n, n, L.bypass_scale, final_name[L.bypass_layer], cur))
            cur = n
        final_name[li] = cur
    cfg.append("output-node name=output input=%s objective=linear" % final_name[len(model.layers) - 1])
    tm, id2pdf, tid_phone = transition_model(num_units)
    out = b"\0B" + tm + tok("<Nnet3>") + b"\n" + ("\n".join(cfg) + "\n\n").encode()
    out += tok("<NumComponents>") + i32(len(comps))
    for name, body in comps:
        out += tok("<ComponentName>") + tok(name) + body
    out += tok("</Nnet3>") + tok("<LeftContext>") + i32(0) + tok("<RightContext>") + i32(0)
    L = model.layers[-1]
    priors = np.exp(-L.post_offset.astype(np.float64)) if L.post_offset is not None else np.zeros(0)
    out += tok("<Priors>") + vec(priors)
    open(path, "wb").write(out)
    return id2pdf, tid_phone
