#!/usr/bin/env python3
"""Turn a gpurun_out/rNN directory (rocprofv3 kernel trace + PMC passes + bench JSON) into the
tracked summaries under profiles/.

    python tools/summarize_profile.py gpurun_out/r01 profiles r01

HBM traffic per launch follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB
and collected in separate passes; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
reads, so the read side is doubled.
"""
import csv
import glob
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("yl::", "")
    return n[:60]


def _agg(dirpath):
    f = Path(dirpath) / "agg.json"  # tools/reduce_pmc.py (run on the GPU box; the raw CSVs are deleted there)
    return json.loads(f.read_text()) if f.exists() else None


def pmc(dirpath, counter):
    a = _agg(dirpath)
    if a is not None:
        return {k: v[counter][0] / v[counter][1] for k, v in a["per_kernel"].items() if counter in v and v[counter][1]}
    out = defaultdict(lambda: [0.0, 0])
    files = sorted(glob.glob(str(Path(dirpath) / "**" / "*counter_collection.csv"), recursive=True),
                   key=lambda f: Path(f).stat().st_mtime)
    for f in files[-1:]:  # the newest run only (gpurun merges into an existing directory)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = out[short(r["Kernel_Name"])]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return {k: v[0] / v[1] for k, v in out.items() if v[1]}


def family(k):
    """rocprofv3 kernel name -> the kernel family names bench.py's roofline uses (yl_engine_op_info)."""
    if k.startswith("conv_kernel<1,"):
        return "conv1x1_mfma"
    if k.startswith("conv_gemm_kernel"):
        return "conv1x1_gemm_mfma"
    if k.startswith("decode_nms_fused"):
        return "decode_nms_fused"
    if k.startswith("conv3_reg_kernel"):
        return "conv3x3_reg_mfma"
    if k.startswith("dwpw_group_kernel"):
        return "dw3x3+pw1x1+1x1_mfma[x3]"
    if k.startswith("conv_kernel<3,"):
        return "conv3x3_mfma+conv3x3s2_mfma"   # one template serves both strides
    if k.startswith("conv_f8_kernel<1,"):
        return "conv1x1_f8_mfma"
    if k.startswith("conv_f8_kernel<3,"):
        return "conv3x3_f8_mfma+conv3x3s2_f8_mfma"
    if k.startswith("bneck2_kernel"):
        return "bottleneck_fused_mfma"
    if k.startswith("quant_f8_kernel"):
        return "quantize_f8"
    if k.startswith("conv3_tile_kernel"):
        return "conv3x3_tile_mfma"
    if k.startswith("dwpw_kernel"):
        return "dw3x3+pw1x1_mfma"
    if k.startswith("stem_kernel"):
        return "stem_conv3x3s2"
    if k.startswith("stem2_kernel"):
        return "stem+conv3x3s2_mfma"
    if k.startswith("c3k2s_kernel"):
        return "c3k2_fused_mfma"
    if k.startswith("dwconv3_kernel"):
        return "dwconv3x3"
    if k.startswith("sppf_pool"):
        return "sppf_maxpool"
    if k.startswith("attention_kernel"):
        return "attention"
    if k.startswith("decode_candidates_sparse"):
        return "decode_candidates_sparse"
    if k.startswith("decode_exact"):
        return "decode_exact"
    if k.startswith("nms_sort"):
        return "nms_sort"
    if k.startswith("nms_mask"):
        return "nms_mask"
    if k.startswith("nms_scan"):
        return "nms_scan"
    return None


def per_launch_table(src, bench):
    """One row per LAUNCH of a step, by position: the per-op table of the un-profiled run (name, family, algorithmic flops / bytes,
    in-sequence HIP-event time) beside THE SAME launch's rocprofv3 kernel-trace duration and its FETCH_SIZE / WRITE_SIZE bytes
    (read side doubled, gfx950).  bench.py's roofline block quotes rows of this table -- traffic and kernel-only duration of exactly
    the launches whose algorithmic work it divides (round 5 divided one family's work by another family mix's duration)."""
    if not (src / "per_op.json").exists():
        return None
    rows = json.load(open(src / "per_op.json"))
    rows = rows["rows"] if isinstance(rows, dict) else rows

    out = [dict(name=r["name"], kernel=r["kernel"], flops=r["flops"], bytes=r["bytes"], us_in_sequence=round(r["ms"] * 1e3, 2)) for r in rows]
    n = len(out)
    note = []
    def program_only(t):  # (older agg.json files still carry the runtime's copy kernels / the counter-clearing launch per step)
        if not t:
            return t
        keep = [i for i, k in enumerate(t["kernels"]) if not k.startswith("zero_i32_kernel") and not k.startswith("__amd_rocclr")]
        return dict(t, kernels=[t["kernels"][i] for i in keep], mean=[t["mean"][i] for i in keep])

    tr = program_only((_agg(src / "trace") or {}).get("trace"))
    if tr and len(tr["mean"]) == n:
        for i in range(n):
            out[i]["us_rocprof"] = round(tr["mean"][i], 2)
            out[i]["kernel_symbol"] = tr["kernels"][i]
        note.append(f"kernel-trace durations: mean over {tr['n_steps']} steps, one batch in flight")
    else:
        note.append(f"kernel trace: steps of {len(tr['mean']) if tr else 0} dispatches do not match the {n} rows of the per-op table")
    for d, c, key, mul in (("pmc_fetch", "FETCH_SIZE", "read_bytes", 2 * 1024.0), ("pmc_write", "WRITE_SIZE", "write_bytes", 1024.0)):
        st = program_only((_agg(src / d) or {}).get("steps", {}).get(c))
        if st and len(st["mean"]) == n:
            for i in range(n):
                out[i][key] = round(st["mean"][i] * mul)
            note.append(f"{c}: mean over {st['n_steps']} steps")
        else:
            note.append(f"{c}: steps of {len(st['mean']) if st else 0} dispatches do not match the {n} rows of the per-op table")
    return {"_config": (bench or {}).get("bench_config"),
            "_source": "rocprofv3 kernel trace (--streams 1) and separate --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, matched to the per-op "
                       "table by launch position within a step (stem launch to stem launch); read side = 2 x FETCH_SIZE (gfx950); " + "; ".join(note),
            "rows": out}


def sq_table(src, top):
    """Issue / wait / LDS / texture-addresser counters per kernel symbol (two --pmc passes, one batch in flight), averaged per launch."""
    names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS",
             "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_ACTIVE_INST_VMEM", "TA_BUSY_avr"]
    vals = {}
    for c in names:
        v = pmc(src / ("pmc_sq1" if names.index(c) < 8 else "pmc_sq2"), c)
        if v:
            vals[c] = v
    if not vals:
        return []
    lines = ["", "## Issue / wait / LDS / texture-addresser counters of the top kernels (per launch; separate `--pmc` passes, one batch in flight)", "",
             "SQ_* cycle counters are in quad-cycles summed over all waves (MI355X_MICROARCH.md): the shares below are of SQ_WAVE_CYCLES --",
             "`wait` = SQ_WAIT_ANY (wave parked on s_waitcnt / barrier), `issue-stall` = SQ_WAIT_INST_ANY, `active` = SQ_ACTIVE_INST_ANY;",
             "`LDS conflict` = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; `TA busy` = TA_BUSY_avr / GRBM_GUI_ACTIVE where both exist.", "",
             "| kernel | waves | wave Mcyc | wait % | issue-stall % | active % | VALU active % | LDS active % | LDS-issue-stall % | VMEM active % | LDS conflict % | VALU insts/wave | LDS insts/wave |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    g = lambda c, k: vals.get(c, {}).get(k)
    for k in top:
        wc, wv = g("SQ_WAVE_CYCLES", k), g("SQ_WAVES", k)
        if not wc or not wv:
            continue
        pct = lambda c: "" if g(c, k) is None else f"{100.0 * g(c, k) / wc:.1f}"
        conf = "" if not g("SQ_LDS_IDX_ACTIVE", k) else f"{100.0 * (g('SQ_LDS_BANK_CONFLICT', k) or 0.0) / g('SQ_LDS_IDX_ACTIVE', k):.1f}"
        per = lambda c: "" if g(c, k) is None else f"{g(c, k) / wv:.0f}"
        lines.append(f"| `{k}` | {wv:.0f} | {wc / 1e6:.2f} | {pct('SQ_WAIT_ANY')} | {pct('SQ_WAIT_INST_ANY')} | {pct('SQ_ACTIVE_INST_ANY')} | {pct('SQ_ACTIVE_INST_VALU')} | "
                     f"{pct('SQ_ACTIVE_INST_LDS')} | {pct('SQ_WAIT_INST_LDS')} | {pct('SQ_ACTIVE_INST_VMEM')} | {conf} | {per('SQ_INSTS_VALU')} | {per('SQ_INSTS_LDS')} |")
    return lines


def main():
    src, dst, tag = Path(sys.argv[1]), Path(sys.argv[2]), sys.argv[3]
    dst.mkdir(parents=True, exist_ok=True)
    stats = sorted(glob.glob(str(src / "trace" / "**" / "*kernel_stats.csv"), recursive=True),
                   key=lambda f: Path(f).stat().st_mtime)[-1:]
    rows = list(csv.DictReader(open(stats[0]))) if stats else []
    if stats:
        shutil.copy(stats[0], dst / f"{tag}_kernel_stats.csv")
    fetch = pmc(src / "pmc_fetch", "FETCH_SIZE")
    write = pmc(src / "pmc_write", "WRITE_SIZE")
    miss = pmc(src / "pmc_miss", "TCC_MISS_sum")          # L2 misses: 128-byte lines fetched from the fabric (independent of FETCH_SIZE)
    rdreq = pmc(src / "pmc_miss", "TCC_EA0_RDREQ_sum")
    rd32 = pmc(src / "pmc_miss", "TCC_EA0_RDREQ_32B_sum")
    mfma_busy = pmc(src / "pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES")
    one_stream = (src / "pmc_mfma.log").exists() and "--streams 1" in (src / "pmc_mfma.log").read_text()[:4000]
    gui = pmc(src / "pmc_mfma", "GRBM_GUI_ACTIVE")
    bench = None
    if (src / "bench.json").exists():
        txt = (src / "bench.json").read_text().strip().splitlines()
        bench = json.loads(txt[-1]) if txt else None
        (dst / f"{tag}_bench.json").write_text(json.dumps(bench, indent=1))
    if (src / "per_op.json").exists():
        shutil.copy(src / "per_op.json", dst / f"{tag}_per_op.json")
    lines = [f"# {tag}: rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline` (one batch in flight: kernel-only durations)", ""]
    if bench:
        lines += [f"bench (un-profiled run): **{bench['value']} {bench['unit']}**, {bench['ms_per_step']} ms/step, "
                  f"roofline {json.dumps(bench.get('roofline'))}", ""]
    lines += ["HBM columns: FETCH_SIZE / WRITE_SIZE (KiB) from separate `--pmc` passes, averaged per launch; the read",
              "side is doubled as MI355X_MICROARCH.md prescribes for gfx950.  Round 3 settles whether that also holds for the 3x3",
              "kernels' tap-shifted loads (round 2: 'probably over-counts by up to 2x'): it does -- see the cross-check columns.", "",
              "MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (1024 x GRBM_GUI_ACTIVE / 8 XCDs), from a",
              "separate pass" + (" with one batch in flight." if one_stream else
              " with 20 batches in flight (round 4: `--streams 1` under counter collection crashed the profiled process): dispatches of "
              "different queues overlap, a dispatch's GRBM_GUI_ACTIVE then spans its neighbours' time too -- read the column as a LOWER "
              "bound, about half of the one-batch figures (round 3, and commit a06b0e0 of round 4).") + ("" if mfma_busy else
              "  NOT in this collection: the profiled process of that pass ended with a GPU memory access fault ~2 s after its "
              "start, before the engine existed (pmc_mfma.log; two of four attempts this round, the other PMC passes never) -- "
              "the column is empty rather than re-tried on a shared node."), "",
              "Read cross-check (round 3, separate pass): TCC_MISS_sum counts the L2 misses of reads AND writes in 128-byte lines --",
              "for every kernel below `misses x 128 B` = `2 x FETCH_SIZE + WRITE_SIZE` within 1-2 % (compare the two read columns with the",
              "write column) and TCC_EA0_RDREQ x 128 B equals 2 x FETCH_SIZE too, none of them 32-byte requests.  So the doubled",
              "FETCH_SIZE IS the read traffic beyond L2 for the tap-shifted 3x3 loads as well."] + ([
              "The first profile of this round (tap-major",
              "K order everywhere) showed model.3 (3x3 stride 2, 64 -> 64 @160x160 -> 80x80, batch 32: `conv_kernel<3, 4, 4, true, true>`)",
              "fetching 225 MB for a 105 MB input = 2.14 x: its nine taps swept a footprint of ~11 MB per XCD against 4 MB of L2, the",
              "re-reads served by the Infinity Cache, which these counters include.  With the plane-major K order that layer is the row",
              "`conv_kernel<3, 4, 4, false, true>` below: ~108 MB."] if tag.startswith("r") else []) + ["",
              "| kernel | calls | avg us | total % | HBM read MB/launch (2 x FETCH_SIZE) | read MB/launch (L2 misses x 128 B) | RDREQ M (32 B share) | HBM write MB/launch | MFMA busy % |",
              "|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        if float(r["Percentage"]) < 0.05:
            continue
        k = short(r["Name"])
        rd = fetch.get(k)
        wr = write.get(k)
        mb = ""
        if k in mfma_busy and gui.get(k):
            mb = f"{100.0 * mfma_busy[k] / (1024.0 * gui[k] / 8.0):.1f}"
        ms_ = "" if k not in miss else f"{miss[k] * 128 / 1e6:.1f}"
        rq = "" if k not in rdreq else f"{rdreq[k] / 1e6:.2f} ({100.0 * rd32.get(k, 0.0) / max(rdreq[k], 1.0):.0f} %)"
        lines.append(f"| `{k}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} | "
                     f"{'' if rd is None else f'{2 * rd * 1024 / 1e6:.1f}'} | {ms_} | {rq} | {'' if wr is None else f'{wr * 1024 / 1e6:.1f}'} | {mb} |")
    top = [short(r["Name"]) for r in sorted(rows, key=lambda r: -float(r["Percentage"]))[:12]]
    lines += sq_table(src, top)
    pl = per_launch_table(src, bench)
    if pl:
        (dst / f"{tag}_per_launch.json").write_text(json.dumps(pl, indent=1))
        lines += ["", "## Per launch of one step (position-matched: per-op table, kernel trace, PMC)", "",
                  "| op | family | alg MB | GFLOP | in-sequence us | rocprof us | PMC read MB | PMC write MB | PMC / alg |", "|---|---|---|---|---|---|---|---|---|"]
        for r in pl["rows"]:
            t = (r.get("read_bytes", 0) + r.get("write_bytes", 0))
            ratio = "" if not t or not r["bytes"] else f"{t / r['bytes']:.2f}"
            lines.append(f"| {r['name'][:60]} | {r['kernel']} | {r['bytes'] / 1e6:.1f} | {r['flops'] / 1e9:.2f} | {r['us_in_sequence']} | {r.get('us_rocprof', '')} | "
                         f"{r.get('read_bytes', 0) / 1e6:.1f} | {r.get('write_bytes', 0) / 1e6:.1f} | {ratio} |")
    (dst / f"{tag}_summary.md").write_text("\n".join(lines) + "\n")
    # per-family HBM bytes per launch (launch-weighted over the template instances), read by bench.py
    fam = {}
    for r in rows:
        k = short(r["Name"])
        f = family(k)
        if f is None or k not in fetch or k not in write:
            continue
        a = fam.setdefault(f, dict(launches=0, read=0.0, write=0.0, ns=0.0))
        n = int(r["Calls"])
        a["launches"] += n
        a["read"] += n * 2 * fetch[k] * 1024
        a["write"] += n * write[k] * 1024
        a["ns"] += n * float(r["AverageNs"])
    traffic = {f: {"read_bytes_per_launch": round(a["read"] / a["launches"]),
                   "write_bytes_per_launch": round(a["write"] / a["launches"]),
                   "avg_launch_us_rocprof": round(a["ns"] / a["launches"] / 1e3, 2),  # kernel-trace duration, launch-weighted
                   "launches_in_trace": a["launches"]} for f, a in fam.items()}
    traffic["_source"] = (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py, {tag}; "
                          "read side = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md)")
    # the configuration these counters belong to: bench.py only quotes a traffic file whose _config equals its own
    traffic["_config"] = (bench or {}).get("bench_config")
    (dst / f"{tag}_traffic.json").write_text(json.dumps(traffic, indent=1))
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main()
