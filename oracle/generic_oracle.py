"""generic_oracle.py -- TEST INFRASTRUCTURE ONLY (never imported by the product).

numpy restatement of ncnn's published layer semantics for the generic graphs the MI355X executor accepts
(upscale_video_amd/csrc/uva_generic.h), i.e. what `-m r` runs in the reference
(/root/reference/models/4x_Valar_v1.param:3-1208 through upscale/upscale_processing.py:913-916):

  Convolution   ncnn convolution.cpp: OIHW weights, stride 1, zero 'same' padding, optional bias,
                activation_type 2 = LeakyReLU(slope)            (.param: 0=cout 1=k 4=pad 5=bias 6=wsize 9=2 -23310=1,slope)
  Concat        axis 0 (channels)                               (:8)
  BinaryOp      op 0 = ADD                                      (:11)
  Eltwise       op 1 = SUM with coefficients                    (:21  0=1 -23301=2,0.2,1.0)
  Interp        resize_type 1 = nearest, integer scale: out[y][x] = in[y // s][x // s]   (:1203)
  Split         aliases;  PReLU per channel;  PixelShuffle mode 0 (PyTorch order)
  .bin          per Convolution: u32 flag (0x01306B47 = fp16 payload padded to 4 bytes, 0 = fp32), weights, then
                fp32 bias if bias_term; per PReLU: fp32 slopes   (ncnn modelbin.cpp)

PARITY UNPINNED: the Valar weights are a missing blob upstream (.MISSING_LARGE_BLOBS) and ncnn is not
installable here, so this file pins nothing against the reference; it is an independent second
implementation the HIP executor is compared with on synthetic weights (and, for the SRVGGNetCompact
graphs with their real weights, it is itself checked against oracle/oracle.c).
"""
import struct

import numpy as np

FP16_FLAG = 0x01306B47


def parse_param(path):
    with open(path) as f:
        toks = f.read().split("\n")
    assert toks[0].strip() == "7767517", "bad magic"
    nl, _ = (int(v) for v in toks[1].split())
    layers = []
    for line in toks[2:]:
        p = line.split()
        if len(p) < 4:
            continue
        typ, name, nin, nout = p[0], p[1], int(p[2]), int(p[3])
        ins, outs = p[4:4 + nin], p[4 + nin:4 + nin + nout]
        kv = {}
        for t in p[4 + nin + nout:]:
            k, v = t.split("=")
            kv[int(k)] = v
        layers.append(dict(type=typ, name=name, ins=ins, outs=outs, kv=kv))
    assert len(layers) == nl
    return layers


def _arr(kv, key):
    if key not in kv:
        return []
    return [float(v) for v in kv[key].split(",")[1:]]


def conv_shapes(layers):
    """[(name, cout, cin, k, has_bias)] in file order, channel counts by shape inference"""
    ch = {}
    out = []
    for L in layers:
        t, kv = L["type"], L["kv"]
        if t == "Input":
            ch[L["outs"][0]] = 3
        elif t == "Split":
            for o in L["outs"]:
                ch[o] = ch[L["ins"][0]]
        elif t == "Convolution":
            cout, k, ws = int(kv[0]), int(kv[1]), int(kv[6])
            cin = ch[L["ins"][0]]
            assert ws == cout * cin * k * k, L["name"]
            out.append((L["name"], cout, cin, k, int(kv.get(5, 0)) != 0))
            ch[L["outs"][0]] = cout
        elif t == "Concat":
            ch[L["outs"][0]] = sum(ch[i] for i in L["ins"])
        elif t == "PixelShuffle":
            f = int(kv.get(0, 1))
            ch[L["outs"][0]] = ch[L["ins"][0]] // (f * f)
        else:
            ch[L["outs"][0]] = ch[L["ins"][0]]
    return out


def write_synthetic_bin(param_path, bin_path, seed=0, fp16=True, gain=0.7):
    """Random weights that keep activations O(1) through hundreds of layers (He-style scaling times `gain`)."""
    rng = np.random.default_rng(seed)
    layers = parse_param(param_path)
    shapes = {n: (co, ci, k, b) for n, co, ci, k, b in conv_shapes(layers)}
    with open(bin_path, "wb") as f:
        for L in layers:
            if L["type"] == "Convolution":
                co, ci, k, has_bias = shapes[L["name"]]
                w = rng.standard_normal(co * ci * k * k).astype(np.float32) * np.float32(gain / np.sqrt(ci * k * k))
                if fp16:
                    f.write(struct.pack("<I", FP16_FLAG))
                    raw = w.astype(np.float16).tobytes()
                    f.write(raw + b"\0" * (-len(raw) % 4))
                else:
                    f.write(struct.pack("<I", 0))
                    f.write(w.tobytes())
                if has_bias:
                    f.write((rng.standard_normal(co).astype(np.float32) * np.float32(0.05)).tobytes())
            elif L["type"] == "PReLU":
                f.write(rng.uniform(0.05, 0.3, int(L["kv"][0])).astype(np.float32).tobytes())


class Model:
    def __init__(self, param_path, bin_path):
        self.layers = parse_param(param_path)
        raw = open(bin_path, "rb").read()
        off = 0
        shapes = {n: (co, ci, k, b) for n, co, ci, k, b in conv_shapes(self.layers)}
        self.w, self.b, self.slopes = {}, {}, {}
        for L in self.layers:
            if L["type"] == "Convolution":
                co, ci, k, has_bias = shapes[L["name"]]
                n = co * ci * k * k
                flag, = struct.unpack_from("<I", raw, off)
                off += 4
                if flag == FP16_FLAG:
                    w = np.frombuffer(raw, np.float16, n, off).astype(np.float32)
                    off += (2 * n + 3) // 4 * 4
                else:
                    assert flag == 0
                    w = np.frombuffer(raw, np.float32, n, off).copy()
                    off += 4 * n
                self.w[L["name"]] = w.reshape(co, ci, k, k)
                if has_bias:
                    self.b[L["name"]] = np.frombuffer(raw, np.float32, co, off).copy()
                    off += 4 * co
                else:
                    self.b[L["name"]] = np.zeros(co, np.float32)
            elif L["type"] == "PReLU":
                n = int(L["kv"][0])
                self.slopes[L["name"]] = np.frombuffer(raw, np.float32, n, off).copy()
                off += 4 * n
        assert off == len(raw), "unread bytes in the .bin"

    @staticmethod
    def _conv(x, w, b, k):
        c, h, wd = x.shape
        if k == 3:
            xp = np.zeros((c, h + 2, wd + 2), np.float32)
            xp[:, 1:-1, 1:-1] = x
            cols = np.stack([xp[:, dy:dy + h, dx:dx + wd] for dy in range(3) for dx in range(3)], axis=1)   # c, 9, h, w
            out = np.tensordot(w.reshape(w.shape[0], c, 9), cols, axes=([1, 2], [0, 1]))
        else:
            out = np.tensordot(w.reshape(w.shape[0], c), x, axes=([1], [0]))
        return (out + b[:, None, None]).astype(np.float32)

    def forward(self, x, f16_storage=False):
        """x: f32 [3][h][w] (normalised) -> f32 [3][h*s][w*s].  f16_storage rounds every blob to fp16 (the
        executor's storage precision; accumulation stays fp32 there and here)."""
        q = (lambda a: a.astype(np.float16).astype(np.float32)) if f16_storage else (lambda a: a)
        blobs = {}
        for L in self.layers:
            t, kv, ins, outs = L["type"], L["kv"], L["ins"], L["outs"]
            if t == "Input":
                blobs[outs[0]] = q(np.asarray(x, np.float32))
            elif t == "Split":
                for o in outs:
                    blobs[o] = blobs[ins[0]]
            elif t == "Convolution":
                w = self.w[L["name"]]
                if f16_storage:
                    w = q(w)
                y = self._conv(blobs[ins[0]], w, self.b[L["name"]], int(kv[1]))
                if int(kv.get(9, 0)) == 2:
                    slope = np.float32(_arr(kv, -23310)[0])
                    y = np.where(y > 0, y, y * slope).astype(np.float32)
                blobs[outs[0]] = q(y)
            elif t == "Concat":
                blobs[outs[0]] = np.concatenate([blobs[i] for i in ins], axis=0)
            elif t == "BinaryOp":
                assert int(kv.get(0, 0)) == 0
                blobs[outs[0]] = q(blobs[ins[0]] + blobs[ins[1]])
            elif t == "Eltwise":
                assert int(kv.get(0, 0)) == 1
                c = _arr(kv, -23301) or [1.0, 1.0]
                blobs[outs[0]] = q(blobs[ins[0]] * np.float32(c[0]) + blobs[ins[1]] * np.float32(c[1]))
            elif t == "Interp":
                s = int(float(kv.get(1, 1.0)))
                blobs[outs[0]] = np.repeat(np.repeat(blobs[ins[0]], s, axis=1), s, axis=2)
            elif t == "PReLU":
                sl = self.slopes[L["name"]][:, None, None]
                a = blobs[ins[0]]
                blobs[outs[0]] = q(np.where(a < 0, a * sl, a).astype(np.float32))
            elif t == "PixelShuffle":
                f = int(kv.get(0, 1))
                a = blobs[ins[0]]
                c, h, w = a.shape
                blobs[outs[0]] = a.reshape(c // (f * f), f, f, h, w).transpose(0, 3, 1, 4, 2).reshape(c // (f * f), h * f, w * f)
            else:
                raise ValueError("layer type " + t)
        return blobs["output"]

    def apply_u8(self, img, f16_storage=False):
        """u8 HWC BGR -> u8 HWC BGR like apply_model / process_tile on one plane (upscale_processing.py:263-288)"""
        x = img.transpose(2, 0, 1).astype(np.float32) * np.float32(1 / 255.0)
        y = self.forward(x, f16_storage).transpose(1, 2, 0) * np.float32(255.0)
        return np.clip(np.rint(y), 0, 255).astype(np.uint8)
