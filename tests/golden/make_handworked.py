"""Hand-worked known-answer fixtures for the pieces whose reference implementation (MONAI) is not importable here.

    python tests/golden/make_handworked.py        # rewrites tests/golden/handworked.json

WHY. DiceLoss / GeneralizedDiceLoss, SlidingWindowInferer and DynUNet live in MONAI (un-vendored, requirements.txt:4 of the
reference); oracle/torch_ops.py, oracle/sliding_window_ref.py and oracle/dynunet_ref.py RESTATE them with torch ops, so a mistake
in the restated formula would be shared by oracle and kernels ("parity unpinned"). The cases below are small enough to be worked by
hand; this script evaluates them from first principles with scalar Python arithmetic only -- no torch, no numpy, none of the oracle's
code -- following the definitions in SURVEY.md 8a (a13), 8f-1 and appendices C / D, and the closed forms noted next to each case
can be checked with a pocket calculator. tests/test_handworked.py runs the oracle AND the HIP kernels against the stored numbers.
It narrows "unpinned" to: "agrees with the published formulas as written here"; it cannot prove agreement with a MONAI release.

CASE dice.   logits chosen so that sigmoid gives exact quarters: z = ln 3 -> p = 3/4, z = -ln 3 -> p = 1/4, z = 0 -> p = 1/2.
  n = 1, c = 2, 4 voxels.  channel 0: p = (3/4, 1/4, 1/2, 1/2), y = (1, 0, 1, 0):  I = 3/4 + 1/2 = 5/4,  sum p = 2,  sum y = 2
      f0 = 1 - (2 * 5/4 + 1e-5) / (2 + 2 + 1e-5) = 1 - 2.50001 / 4.00001 = 0.374999062502...
  channel 1 (EMPTY class): p = (1/4, 1/4, 1/4, 1/4), y = 0:  I = 0, sum p = 1, sum y = 0
      f1 = 1 - 1e-5 / 1.00001 = 0.999990000099999
  DiceLoss(sigmoid=True) = (f0 + f1) / 2 = 0.68749453130...;  squared_pred: denominators use sum p^2: ch0 9/16+1/16+1/4+1/4 = 9/8,
  ch1 4/16 = 1/4.
CASE gdl.    GeneralizedDiceLoss(w_type="square", sigmoid): per sample w_c = 1 / (sum_v y_c)^2, an EMPTY class has w = inf which MONAI
  replaces by the largest finite weight of that sample; loss = 1 - (2 * sum_c w_c I_c + 1e-5) / (sum_c w_c (sum p_c + sum y_c) + 1e-5).
  Same tensors: w0 = 1/4, w1 = inf -> 1/4:  numerator 2 * (1/4 * 5/4 + 1/4 * 0) = 5/8, denominator 1/4 * 4 + 1/4 * 1 = 5/4
      loss = 1 - (0.625 + 1e-5) / (1.25 + 1e-5) = 0.499996000032...
CASE dice options (monai DiceLoss kwargs beyond the shipped configuration), on the two-sample tensors of CASE dice:
  jaccard: the denominator D = sum p + sum y becomes 2 (D - I): sample 0 channel 0: 1 - 2.50001 / (2 (4 - 5/4) + 1e-5) = 1 - 2.50001 / 5.50001.
  weight (w0, w1) multiplies the per-class values before the reduction; reduction "sum" adds the n * c values, "none" returns them.
  softmax (instead of sigmoid) over the TWO channels of a voxel: (z0, z1) = (ln 3, -ln 3) -> p = (9/10, 1/10); (-ln 3, -ln 3) -> (1/2, 1/2);
  (0, -ln 3) -> (3/4, 1/4). Sample 0: p0 = (9/10, 1/2, 3/4, 3/4), y0 = (1, 0, 1, 0): I = 33/20, sum p = 29/10, sum y = 2:
      f = 1 - (3.3 + 1e-5) / (4.9 + 1e-5) = 0.32653...
  to_onehot_y: the target is a label map l(v) in {0, 1}; y_c(v) = [l(v) = c]. include_background=False keeps channel 1 only (softmax
  still runs over both channels first).
CASE window. 1-D plan of MONAI's dense_patch_slices: image 20, roi 8, overlap 0.25 -> interval int(8 * 0.75) = 6; starts
  min(i * 6, 20 - 8) until the window reaches the end: (0, 6, 12).  image 19 -> (0, 6, 11).  image 8 -> (0).  overlap 0.5 on 240 with
  roi 128: interval 64 -> (0, 64, 112).  Gaussian importance, roi 8, sigma = 0.125 * 8 = 1, centre (8 - 1) / 2 = 3.5:
  g_i = exp(-(i - 3.5)^2 / 2);  divided by the maximum (g_3 = g_4): w = (e^-6, e^-3, e^-1, 1, 1, e^-1, e^-3, e^-6) =
  (0.00247875, 0.04978707, 0.36787944, 1, ...). The 3-D map is the outer product of three such vectors divided by its maximum and
  THEN clamped at 1e-3: in 1-D the clamp is inactive for roi 8 (e^-6 > 1e-3), in 3-D it is not (e^-6 * e^-1 < 1e-3 already one
  voxel off the in-plane centre), so the map is not separable. A constant rescaling of the map cancels in sum(w * pred) / sum(w):
  MONAI versions that do or do not divide by the maximum give the same output wherever the clamp is inactive.
  Inference: volume (20, 8, 8) with one channel holding v(z) = z, roi (8, 8, 8), overlap 0.25, predictor(window) = window + 100 * k
  for the k-th window (k = 0, 1, 2 in plan order): out(z, y, x) = z + 100 * sum_k k * w3(z - s_k, y, x) / sum_k w3(z - s_k, y, x),
  w3(a, y, x) = max(g_a g_y g_x, 1e-3). At the in-plane centre (y, x) = (3, 3) the clamp is inactive: z = 7 lies in windows 0
  (local 7) and 1 (local 1): out = 7 + 100 * e^-3 / (e^-6 + e^-3) = 7 + 100 / (1 + e^-3) = 102.2574...; in the corner (0, 0)
  (in-plane factor e^-12) every weight is the clamp value 1e-3 and the windows average plainly: out(7, 0, 0) = 7 + 100 * 1/2 = 57.
CASE dynunet. MONAI DynUNet(spatial_dims 3, in 1, out 2, kernel 3, strides (1, 2), upsample kernel 2, filters (4, 4)), appendix D:
  input_block = [Conv3d(1->4, k3, s1, p1, no bias) -> InstanceNorm3d(affine) -> LeakyReLU(0.01)] x 2 (second conv 4->4),
  bottleneck  = the same block with stride 2 in its first conv (4->4), upsample = ConvTranspose3d(4->4, k2, s2, no bias) ->
  cat((up, skip), 1) -> block(8->4), output = Conv3d(4->2, k1, bias). Weights are small integers divided by 8 (exact in fp32),
  written out below; the evaluation is plain nested loops over (co, ci, taps, voxels) exactly as appendix C defines the ops
  (cross-correlation with zero padding; biased variance, eps 1e-5; transposed conv o = 2 i + t).
"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LN3 = math.log(3.0)
EPS = 1e-5


def sigmoid(z):
    return 1.0 / (1.0 + math.exp(-z))


# ---- Dice / GeneralizedDice ------------------------------------------------------------------------
def dice_cases():
    logits = [[[LN3, -LN3, 0.0, 0.0], [-LN3, -LN3, -LN3, -LN3]]]          # [n][c][v]
    target = [[[1, 0, 1, 0], [0, 0, 0, 0]]]
    # a second sample for the batch=True form
    logits2 = logits + [[[0.0, LN3, LN3, -LN3], [LN3, 0.0, -LN3, 0.0]]]
    target2 = target + [[[0, 1, 1, 0], [1, 1, 0, 0]]]

    def sums(lg, tg, squared):
        out = []
        for n in range(len(lg)):
            row = []
            for c in range(len(lg[n])):
                p = [sigmoid(z) for z in lg[n][c]]
                y = tg[n][c]
                inter = sum(pi * yi for pi, yi in zip(p, y))
                sp = sum(pi * pi for pi in p) if squared else sum(p)
                sy = sum(yi * yi for yi in y) if squared else sum(y)
                row.append((inter, sp, sy))
            out.append(row)
        return out

    def dice(lg, tg, squared=False, batch=False):
        s = sums(lg, tg, squared)
        n, c = len(s), len(s[0])
        if batch:
            f = []
            for ch in range(c):
                inter = sum(s[k][ch][0] for k in range(n))
                den = sum(s[k][ch][1] + s[k][ch][2] for k in range(n))
                f.append(1.0 - (2.0 * inter + EPS) / (den + EPS))
            return sum(f) / c
        f = [1.0 - (2.0 * s[k][ch][0] + EPS) / (s[k][ch][1] + s[k][ch][2] + EPS) for k in range(n) for ch in range(c)]
        return sum(f) / (n * c)

    def gdl(lg, tg):
        s = sums(lg, tg, False)
        losses = []
        for k in range(len(s)):
            w = []
            for inter, sp, sy in s[k]:
                w.append(math.inf if sy == 0 else 1.0 / (sy * sy))
            finite = [v for v in w if v != math.inf]
            mx = max(finite) if finite else 0.0
            w = [mx if v == math.inf else v for v in w]
            num = 2.0 * sum(wi * si[0] for wi, si in zip(w, s[k])) + EPS
            den = sum(wi * (si[1] + si[2]) for wi, si in zip(w, s[k])) + EPS
            losses.append(1.0 - num / den)
        return sum(losses) / len(losses)

    return {"logits": logits, "target": target, "logits2": logits2, "target2": target2, "shape": [1, 2, 1, 2, 2],
            "dice": dice(logits, target), "dice_squared_pred": dice(logits, target, squared=True),
            "dice_two_samples": dice(logits2, target2), "dice_two_samples_batch": dice(logits2, target2, batch=True),
            "gdl": gdl(logits, target), "gdl_two_samples": gdl(logits2, target2),
            "closed_forms": {"f0": 1 - 2.50001 / 4.00001, "f1": 1 - 1e-5 / 1.00001, "gdl": 1 - 0.62501 / 1.25001}}


def dice_option_cases(base):
    """The per-(n, c) Dice values of the two-sample tensors under the option combinations, from scalar arithmetic."""
    lg, tg = base["logits2"], base["target2"]
    labels = [[1 if tg[n][1][v] else 0 for v in range(4)] for n in range(2)]      # a label map: class 1 where channel 1 of the target is set

    def probs(n, v, act):
        z = [lg[n][c][v] for c in range(2)]
        if act == "sigmoid":
            return [sigmoid(t) for t in z]
        m = max(z)
        e = [math.exp(t - m) for t in z]
        return [t / sum(e) for t in e]

    def terms(act="sigmoid", onehot=False, jaccard=False, squared=False, batch=False, weight=None, background=True):
        c0 = 0 if background else 1
        s = [[[0.0, 0.0, 0.0] for _ in range(2)] for _ in range(2)]
        for n in range(2):
            for v in range(4):
                p = probs(n, v, act)
                for c in range(2):
                    y = (1.0 if labels[n][v] == c else 0.0) if onehot else float(tg[n][c][v])
                    s[n][c][0] += p[c] * y
                    s[n][c][1] += p[c] * p[c] if squared else p[c]
                    s[n][c][2] += y * y if squared else y
        rows = [[[sum(s[n][c][k] for n in range(2)) for k in range(3)] for c in range(2)]] if batch else s
        out = []
        for row in rows:
            r = []
            for c in range(c0, 2):
                inter, den = row[c][0], row[c][1] + row[c][2]
                if jaccard:
                    den = 2.0 * (den - inter)
                f = 1.0 - (2.0 * inter + EPS) / (den + EPS)
                if weight is not None and 2 - c0 != 1:
                    f *= weight[c - c0]
                r.append(f)
            out.append(r)
        return out

    def flat(t):
        return [x for r in t for x in r]

    def mean(t):
        return sum(flat(t)) / len(flat(t))

    return {"labels": labels,
            "jaccard": mean(terms(jaccard=True)), "jaccard_squared_batch": mean(terms(jaccard=True, squared=True, batch=True)),
            "weight": mean(terms(weight=[2.0, 0.5])), "sum": sum(flat(terms())), "none": terms(), "none_batch": terms(batch=True),
            "softmax": mean(terms(act="softmax")), "softmax_none": terms(act="softmax"),
            "softmax_onehot": mean(terms(act="softmax", onehot=True)),
            "softmax_onehot_nobg_sum": sum(flat(terms(act="softmax", onehot=True, background=False))),
            "sigmoid_onehot_jaccard_weight_none": terms(onehot=True, jaccard=True, weight=[2.0, 0.5]),
            "closed_forms": {"jaccard_n0c0": 1 - 2.50001 / 5.50001, "softmax_n0c0": 1 - 3.30001 / 4.90001}}


# ---- sliding window -----------------------------------------------------------------------------------------------
def starts_1d(size, roi, overlap):
    if roi == size:
        return [0]
    iv = int(roi * (1.0 - overlap))
    iv = iv if iv > 0 else 1
    out, i = [], 0
    while True:
        s = min(i * iv, size - roi)
        out.append(s)
        if i * iv + roi >= size:
            break
        i += 1
    return out


def gauss_1d(roi, sigma_scale=0.125):
    c = (roi - 1) / 2.0
    g = [math.exp(-0.5 * ((i - c) / (sigma_scale * roi)) ** 2) for i in range(roi)]
    m = max(g)
    return [max(v / m, 1e-3) for v in g]


def window_cases():
    w = gauss_1d(8)
    st = starts_1d(20, 8, 0.25)

    def ramp(y, x):
        out = []
        for z in range(20):
            num = den = 0.0
            for k, s in enumerate(st):
                if s <= z < s + 8:
                    w3 = max(w[z - s] * w[y] * w[x], 1e-3)            # outer product (already max-normalised), then the clamp
                    num += w3 * (z + 100.0 * k)
                    den += w3
            out.append(num / den)
        return out
    out, out_corner, out_off = ramp(3, 3), ramp(0, 0), ramp(3, 5)
    return {"starts": {"20/8/0.25": st, "19/8/0.25": starts_1d(19, 8, 0.25), "8/8/0.25": starts_1d(8, 8, 0.25),
                       "240/128/0.5": starts_1d(240, 128, 0.5), "155/128/0.5": starts_1d(155, 128, 0.5),
                       "240/128/0.25": starts_1d(240, 128, 0.25)},
            "gauss_roi8": w, "gauss_roi8_closed_form": [math.exp(-6), math.exp(-3), math.exp(-1), 1.0, 1.0, math.exp(-1), math.exp(-3), math.exp(-6)],
            "ramp_inference": {"size": [20, 8, 8], "roi": [8, 8, 8], "overlap": 0.25, "out_of_z": out, "out_of_z_corner": out_corner,
                               "out_of_z_y3x5": out_off, "closed_form_z7": 7 + 100.0 / (1 + math.exp(-3)), "closed_form_z7_corner": 57.0}}


# ---- DynUNet, two levels ----------------------------------------------------------------------------------
def zeros(*shape):
    if len(shape) == 1:
        return [0.0] * shape[0]
    return [zeros(*shape[1:]) for _ in range(shape[0])]


def conv3d(x, w, stride, pad):
    """x [ci][D][H][W], w [co][ci][k][k][k] -> [co][Do][Ho][Wo]: y[co,q] = sum w[co,ci,t] x[ci, s q + t - p] (zero padding)."""
    ci_n, D, H, W = len(x), len(x[0]), len(x[0][0]), len(x[0][0][0])
    k = len(w[0][0])
    Do, Ho, Wo = [(s + 2 * pad - k) // stride + 1 for s in (D, H, W)]
    y = zeros(len(w), Do, Ho, Wo)
    for co in range(len(w)):
        for z in range(Do):
            for yy in range(Ho):
                for xx in range(Wo):
                    acc = 0.0
                    for ci in range(ci_n):
                        for a in range(k):
                            iz = stride * z + a - pad
                            if iz < 0 or iz >= D:
                                continue
                            for b in range(k):
                                iy = stride * yy + b - pad
                                if iy < 0 or iy >= H:
                                    continue
                                for c in range(k):
                                    ix = stride * xx + c - pad
                                    if 0 <= ix < W:
                                        acc += w[co][ci][a][b][c] * x[ci][iz][iy][ix]
                    y[co][z][yy][xx] = acc
    return y


def inorm_lrelu(x, gamma, beta, slope=0.01):
    out = []
    for c, ch in enumerate(x):
        vals = [v for pl in ch for row in pl for v in row]
        m = sum(vals) / len(vals)
        var = sum((v - m) ** 2 for v in vals) / len(vals)            # biased
        r = 1.0 / math.sqrt(var + EPS)
        o = [[[(v - m) * r * gamma[c] + beta[c] for v in row] for row in pl] for pl in ch]
        out.append([[[v if v > 0 else slope * v for v in row] for row in pl] for pl in o])
    return out


def tconv2(x, w):
    """ConvTranspose3d(k2, s2, no bias): w [ci][co][2][2][2]; y[co, 2i + t] = sum_ci x[ci, i] w[ci, co, t]."""
    ci_n, D, H, W = len(x), len(x[0]), len(x[0][0]), len(x[0][0][0])
    co_n = len(w[0])
    y = zeros(co_n, 2 * D, 2 * H, 2 * W)
    for ci in range(ci_n):
        for co in range(co_n):
            for z in range(D):
                for yy in range(H):
                    for xx in range(W):
                        v = x[ci][z][yy][xx]
                        for a in range(2):
                            for b in range(2):
                                for c in range(2):
                                    y[co][2 * z + a][2 * yy + b][2 * xx + c] += v * w[ci][co][a][b][c]
    return y


def int_weight(co, ci, k, seed):
    """Deterministic small integers / 8 (exact in fp32): ((7 a + 3 b + 5 c + 11 co + 13 ci + seed) mod 7 - 3) / 8."""
    return [[[[[(((7 * a + 3 * b + 5 * c + 11 * o + 13 * i + seed) % 7) - 3) / 8.0 for c in range(k)] for b in range(k)] for a in range(k)]
             for i in range(ci)] for o in range(co)]


def dynunet_case():
    D = 4
    x = [[[[float(((3 * z + 5 * y + 7 * xx) % 11) - 5) for xx in range(D)] for y in range(D)] for z in range(D)]]     # [1][4][4][4]
    P = {
        "input_block.conv1.conv.weight": int_weight(4, 1, 3, 1), "input_block.conv2.conv.weight": int_weight(4, 4, 3, 2),
        "bottleneck.conv1.conv.weight": int_weight(4, 4, 3, 3), "bottleneck.conv2.conv.weight": int_weight(4, 4, 3, 4),
        "upsamples.0.conv_block.conv1.conv.weight": int_weight(4, 8, 3, 5), "upsamples.0.conv_block.conv2.conv.weight": int_weight(4, 4, 3, 6),
        "output_block.conv.conv.weight": int_weight(2, 4, 1, 7), "output_block.conv.conv.bias": [0.25, -0.5],
    }
    tw = int_weight(4, 4, 2, 8)                                      # indexed [ci][co][2][2][2] below
    P["upsamples.0.transp_conv.conv.weight"] = tw
    names = ["input_block", "bottleneck", "upsamples.0.conv_block"]
    for nm in names:
        for j, (ga, be) in enumerate((([1.0, 0.5, 1.5, 1.0], [0.0, 0.25, -0.25, 0.5]), ([0.75, 1.0, 1.25, 0.5], [0.125, 0.0, -0.125, 0.25])), 1):
            P[f"{nm}.norm{j}.weight"], P[f"{nm}.norm{j}.bias"] = ga, be

    def block(t, nm, stride):
        t = inorm_lrelu(conv3d(t, P[f"{nm}.conv1.conv.weight"], stride, 1), P[f"{nm}.norm1.weight"], P[f"{nm}.norm1.bias"])
        return inorm_lrelu(conv3d(t, P[f"{nm}.conv2.conv.weight"], 1, 1), P[f"{nm}.norm2.weight"], P[f"{nm}.norm2.bias"])
    x0 = block(x, "input_block", 1)
    b = block(x0, "bottleneck", 2)
    up = tconv2(b, tw)
    u0 = block(up + x0, "upsamples.0.conv_block", 1)                  # cat((up, skip), 1): up-sampled channels first
    logits = conv3d(u0, P["output_block.conv.conv.weight"], 1, 0)
    logits = [[[[v + P["output_block.conv.conv.bias"][c] for v in row] for row in pl] for pl in logits[c]] for c in range(2)]
    return {"kwargs": dict(spatial_dims=3, in_channels=1, out_channels=2, kernel_size=[3, 3], strides=[1, 2], upsample_kernel_size=[2],
                           filters=[4, 4]),
            "x": [x], "state_dict": P, "logits": [logits], "skip_x0_sample": x0[0][0][0], "bottleneck_sample": b[0][0][0]}


def main():
    out = {"dice": dice_cases(), "window": window_cases(), "dynunet": dynunet_case()}
    out["dice_options"] = dice_option_cases(out["dice"])
    with open(os.path.join(HERE, "handworked.json"), "w") as f:
        json.dump(out, f)
    d = out["dice"]
    print("dice", d["dice"], "closed form", (d["closed_forms"]["f0"] + d["closed_forms"]["f1"]) / 2)
    print("gdl", d["gdl"], "closed form", d["closed_forms"]["gdl"])
    print("window starts", out["window"]["starts"], "z=7:", out["window"]["ramp_inference"]["out_of_z"][7], out["window"]["ramp_inference"]["closed_form_z7"])


if __name__ == "__main__":
    main()
