"""Every oracle comparison of the -m gpu tests leaves its MEASURED distance here, next to the bar it was held to
("state the parity slack as measured + margin, and show it").

    from parity_report import check_u8, check_f32

    check_u8("tiled 2x 70x75 t32", got, want, vs="fp32 oracle", max_lsb=2, min_psnr=50)

records max |diff| (LSB), PSNR (dB) and the share of differing samples, THEN asserts the bars.  conftest.py writes the
records to gpurun_out/parity_report.json at the end of the session and prints one line per record in pytest's terminal
summary (so that the driver's `-q` log shows them); tools/parity_slack.py turns a report into tests/golden/parity_slack.json,
the committed measurements the product-mode bars are derived from (bar = measured maximum + margin, per model and route).
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = []
_SLACK = None


def psnr_u8(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float((d * d).mean()) if d.size else 0.0
    return 99.0 if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse))


def slack(model, route, what, fallback):
    """Bar for `what` ("u8_differ_share", "layer_rel", "f32_abs") of `model` on `route` from the committed sweep:
    measured maximum + margin (tests/golden/parity_slack.json, written by tools/parity_slack.py from a GPU run's report);
    `fallback` where the file has no entry (a new model / route: the round number it replaces)."""
    global _SLACK
    if _SLACK is None:
        try:
            _SLACK = json.load(open(os.path.join(ROOT, "tests", "golden", "parity_slack.json")))
        except (OSError, ValueError):
            _SLACK = {}
    e = _SLACK.get("bars", {}).get(f"{model}/{route}/{what}")
    return float(e["bar"]) if e else fallback


def fp32_bar(model, route):
    """Bars of a comparison with the fp32 oracle (kernel-independent) as keyword arguments of check_u8: the measured worst
    case of `model` on `route` -/+ a margin (tests/golden/parity_slack.json "fp32_bars", tools/parity_slack.py), never looser
    than the fixed bars they replace; those fixed bars where nothing has been measured yet."""
    slack("", "", "", 0)                                    # loads the file
    e = _SLACK.get("fp32_bars", {}).get(f"{model}/{route}")
    chain = model == "chain"
    if not e:
        return {"max_lsb": 3 if chain else 2, "min_psnr": 48.0 if chain else 50.0}
    return {"max_lsb": int(e["max_lsb"]), "min_psnr": float(e["min_psnr_db"]), "max_share": float(e["max_differ_share"])}


def record(name, **fields):
    rec = {"name": name}
    rec.update({k: (round(v, 6) if isinstance(v, float) else v) for k, v in fields.items()})
    RECORDS.append(rec)
    return rec


STRUCTURE_MIN = 48          # lines shorter than this many pixels say nothing about structure
STRUCTURE_Z = 12.0          # bar on the line statistic below (calibration: profiles/r06_structure_calibration.txt)
STRUCTURE_HALF = 12         # the local baseline of a line = median over this many lines either side


def _line_excess(line_mean, n, frame_mean):
    """How far each line's (row's / column's) mean |diff| stands above its neighbourhood, in units of the standard error
    a line of n independent 0/1 samples would have: z = (c_j - median(c_{j-12..j+12})) / sqrt(max(baseline, frame mean, 1/n) / n).
    The running median follows what the picture does to the error rate (it varies smoothly with the content); a schedule
    bug does not: one wrong column or row, a plane edge, every 74th column."""
    k = len(line_mean)
    if k < 2 * STRUCTURE_HALF + 1:
        return np.zeros(k)
    pad = np.pad(line_mean, STRUCTURE_HALF, mode="reflect")
    win = np.lib.stride_tricks.sliding_window_view(pad, 2 * STRUCTURE_HALF + 1)
    base = np.median(win, axis=1)
    return (line_mean - base) / np.sqrt(np.maximum(np.maximum(base, frame_mean), 1.0 / n) / n)


def structure_u8(d):
    """|diff| [h][w][c] -> {"col_z", "row_z", "col_at", "row_at", "col_ratio", "row_ratio"} or None for small frames.
    No spatial structure means: no single row or column carries visibly more error than its neighbours."""
    if d.ndim != 3 or d.shape[0] < STRUCTURE_MIN or d.shape[1] < STRUCTURE_MIN:
        return None
    m = float(d.mean())
    col = d.mean(axis=(0, 2), dtype=np.float64)
    row = d.mean(axis=(1, 2), dtype=np.float64)
    zc = _line_excess(col, d.shape[0] * d.shape[2], m)
    zr = _line_excess(row, d.shape[1] * d.shape[2], m)
    floor = max(m, 1e-9)
    return {"col_z": float(zc.max()), "col_at": int(zc.argmax()), "row_z": float(zr.max()), "row_at": int(zr.argmax()),
            "col_ratio": float(col.max() / floor), "row_ratio": float(row.max() / floor)}


def check_u8(name, got, want, vs, max_lsb, min_psnr=None, max_share=None, model=None, route=None, structure=True):
    """u8 frames: record, then hold to the bars (max |diff| in LSB, PSNR in dB, share of samples that differ) and -- on
    frames of at least 48 x 48 -- to "the error has no spatial structure" (structure_u8: a row or column whose mean |diff|
    stands STRUCTURE_Z standard errors above its neighbours fails, whatever the LSB and dB bars say: round 5's folded-strip
    bug was one wrong column in 74 inside <= 2 LSB / >= 50 dB)."""
    assert got.shape == want.shape and got.dtype == np.uint8, (name, got.shape, want.shape, got.dtype)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    worst = int(d.max()) if d.size else 0
    share = float((d > 0).mean()) if d.size else 0.0
    p = psnr_u8(got, want)
    st = structure_u8(d) if structure else None
    record(name, kind="u8", vs=vs, model=model, route=route, samples=int(d.size), max_lsb=worst, psnr_db=p, differ_share=share,
           bar_max_lsb=max_lsb, bar_min_psnr_db=min_psnr, bar_max_share=max_share, structure=st,
           bar_structure_z=(STRUCTURE_Z if st else None))
    if st:
        assert st["col_z"] <= STRUCTURE_Z, (name, vs, "structured error: column", st["col_at"], "stands", st["col_z"], "s.e. above its neighbours", st)
        assert st["row_z"] <= STRUCTURE_Z, (name, vs, "structured error: row", st["row_at"], "stands", st["row_z"], "s.e. above its neighbours", st)
    assert worst <= max_lsb, (name, vs, "max |diff|", worst, "bar", max_lsb)
    if min_psnr is not None:
        assert p >= min_psnr, (name, vs, "PSNR", p, "bar", min_psnr)
    if max_share is not None:
        assert share <= max_share, (name, vs, "differing share", share, "bar", max_share)
    return worst, p, share


def check_f32(name, got, want, vs, max_abs, model=None, route=None, what="f32_abs", scale=None):
    """float blobs: record max |diff| (and, with `scale`, relative to the blob's range), then hold to max_abs"""
    err = float(np.abs(got - want).max()) if got.size else 0.0
    record(name, kind="f32", vs=vs, model=model, route=route, what=what, max_abs_err=err,
           rel_to_range=(err / scale if scale else None), bar_max_abs=float(max_abs))
    assert err <= max_abs, (name, vs, "max |err|", err, "bar", max_abs)
    return err


def write(path=None):
    if not RECORDS:
        return None
    out_dir = os.path.join(ROOT, "gpurun_out")
    path = path or os.path.join(out_dir, "parity_report.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"records": RECORDS}, f, indent=1)
    except OSError:
        return None
    return path


def summary_lines():
    for r in RECORDS:
        if r.get("kind") == "u8":
            st = r.get("structure")
            yield ("parity %-58s vs %-34s max %d LSB (bar %s)  PSNR %6.2f dB (bar %s)  differ %.3f %% (bar %s)%s" % (
                r["name"][:58], r["vs"][:34], r["max_lsb"], r["bar_max_lsb"], r["psnr_db"], r["bar_min_psnr_db"],
                100 * r["differ_share"], ("%.1f %%" % (100 * r["bar_max_share"])) if r["bar_max_share"] is not None else "-",
                ("  structure z col %.1f row %.1f (bar %.0f)" % (st["col_z"], st["row_z"], r["bar_structure_z"])) if st else ""))
        else:
            yield ("parity %-58s vs %-34s max |err| %.3e (bar %.3e)%s" % (
                r["name"][:58], r["vs"][:34], r["max_abs_err"], r["bar_max_abs"],
                ("  = %.2e of the range" % r["rel_to_range"]) if r.get("rel_to_range") else ""))
