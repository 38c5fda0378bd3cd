#!/usr/bin/env python3
"""Turn one gpurun_out/<tag>/ measurement session (tools/gpu_measure.sh) into the files kept under profiles/.

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_pmc_traffic.csv    per kernel: launches, mean FETCH_SIZE / WRITE_SIZE (KB, raw counter) and the
                                    HBM bytes per launch after the gfx950 correction (FETCH_SIZE x 2 for wide
                                    streaming reads, MI355X_MICROARCH.md "HBM traffic from rocprofv3")
  profiles/<tag>_bench.json         the bench line of the same session
  profiles/<tag>_summary.md         the numbers side by side (bench roofline vs rocprof average, traffic vs algorithmic)

Usage: python tools/summarize_profiles.py r01b
"""
import csv
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name.split("(")[0]


def pmc_means(path: Path, counter: str):
    acc = defaultdict(lambda: [0, 0.0])
    if not path.exists():
        return {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = acc[short(row["Kernel_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return {k: (n, s / n) for k, (n, s) in acc.items() if n}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = ROOT / "gpurun_out" / tag
    dst = ROOT / "profiles"
    dst.mkdir(exist_ok=True)
    stats = src / "prof" / "ktrace_kernel_stats.csv"
    shutil.copy(stats, dst / f"{tag}_kernel_stats.csv")
    pf_stats = src / "prof_prefill" / "pf_kernel_stats.csv"     # the prompt pass alone (tools/prefill_bench.py)
    if pf_stats.exists():
        shutil.copy(pf_stats, dst / f"{tag}_prefill_kernel_stats.csv")
    bench = json.loads((src / "bench.json").read_text().strip().splitlines()[-1])
    (dst / f"{tag}_bench.json").write_text(json.dumps(bench) + "\n")

    fetch = pmc_means(src / "pmc_fetch" / "f_counter_collection.csv", "FETCH_SIZE")
    write = pmc_means(src / "pmc_write" / "w_counter_collection.csv", "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        nf, f_kb = fetch.get(k, (0, 0.0))
        nw, w_kb = write.get(k, (0, 0.0))
        rows.append((k, nf, f_kb, nw, w_kb, f_kb * 1024 * 2 + w_kb * 1024))
    with open(dst / f"{tag}_pmc_traffic.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches_fetch_pass", "FETCH_SIZE_KB_mean_raw", "launches_write_pass", "WRITE_SIZE_KB_mean_raw",
                    "hbm_bytes_per_launch_corrected"])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]:.1f}", r[3], f"{r[4]:.1f}", f"{r[5]:.0f}"])

    # which commit the PMC table was collected at: bench.py shows it next to roofline.traffic (a stale table is then visible)
    import datetime
    import subprocess
    try:
        commit = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
    except Exception:
        commit = None
    (dst / f"{tag}_meta.json").write_text(json.dumps({"commit": commit, "collected": datetime.date.today().isoformat(),
                                                       "command": "tools/gpu_measure.sh " + tag}) + "\n")

    rl = bench["roofline"]
    kname = rl["kernel"]
    if kname.endswith(">") and kname not in {short(k) for k in []}:
        pass
    if rl.get("traffic") is None:   # the bench line ran before this session's PMC table existed: fill it from the same session
        for r in rows:
            if r[0] == kname and r[1] > 0:
                rl["traffic"], rl["traffic_source"] = int(r[5]), f"{tag}_pmc_traffic.csv"
        (dst / f"{tag}_bench.json").write_text(json.dumps(bench) + "\n")
    with open(stats, newline="") as f:
        st = {short(r["Name"]): r for r in csv.DictReader(f)}
    if kname not in st:   # a bench line written with the symbol name of an earlier template signature (mvq_kernel<1u> -> <1u, false>)
        alt = kname[:-1] + ", false>"
        if alt in st:
            for d in (bench.get("kernels", {}),):
                for k in list(d):
                    if k.endswith(">") and k[:-1] + ", false>" in st:
                        d[k[:-1] + ", false>"] = d.pop(k)
            kname = rl["kernel"] = alt
    lines = [f"# {tag}: measurement summary", "",
             f"bench: {bench['value']} {bench['unit']}, {bench['ms_per_step']} ms/token, n_gpus={bench['n_gpus']}", "",
             "| kernel | rocprof calls | rocprof avg us | share % | bench avg us (hipEvent, bracket removed, + dispatch gap) | FETCH raw KB | WRITE raw KB | HBM bytes/launch (corrected) |",
             "|---|---|---|---|---|---|---|---|"]
    traffic = {r[0]: r for r in rows}
    for k, r in st.items():
        if not k.startswith("lgh::"):
            continue
        b = bench.get("kernels", {}).get(k, {})
        bavg = ""
        if b:
            bavg = f"{b['avg_us'] - rl.get('event_bracket_us', 0.0) + rl.get('dispatch_gap_us', 0.0):.2f}"
        t = traffic.get(k)
        lines.append(f"| `{k}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} | {bavg} | "
                     f"{t[2]:.0f} | {t[4]:.0f} | {t[5] / 1e6:.2f} MB |" if t else
                     f"| `{k}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} | {bavg} | | | |")
    lines += ["",
              f"dominant kernel `{kname}`: algorithmic {rl['alg_bytes_per_launch'] / 1e6:.2f} MB/launch, "
              f"bench avg {rl['avg_launch_us']} us -> {rl['achieved']} GB/s = {rl['frac']:.3f} of {rl['peak']} GB/s; "
              f"rocprof avg {float(st[kname]['AverageNs']) / 1e3:.2f} us."]
    if kname in traffic:
        lines.append(f"PMC traffic for it: {traffic[kname][5] / 1e6:.2f} MB/launch "
                     f"(FETCH_SIZE x2 + WRITE_SIZE; the PMC passes ran `bench.py --steps 8 --warmup 2 --prompt 8`).")
    if bench.get("prefill"):
        pf = bench["prefill"]
        lines += ["", f"prompt pass ({pf['tokens']} tokens): lgh_prefill_batch {pf['forward_batch_ms']} ms = {pf['forward_batch_tokens_per_s']} tokens/s "
                      f"(batched={pf['batched']}), token by token {pf['token_by_token_tokens_per_s']} tokens/s"
                      + (f"; {pf['roofline']['achieved']} TFLOP/s = {pf['roofline']['frac']:.3f} of the {pf['roofline']['peak']} TFLOP/s dense f16 peak" if pf.get("roofline") else "")]
        log = src / "prefill_bench.log"
        if log.exists():
            lines += [l.strip() for l in log.read_text().splitlines() if "tokens/s" in l]
        if pf_stats.exists():
            with open(pf_stats, newline="") as f:
                rows_pf = [r for r in csv.DictReader(f) if "pf_" in r["Name"] or "attn_partial" in r["Name"] or "embed_batch" in r["Name"]]
            lines += ["", "| prompt-pass kernel (6 passes of 128 tokens) | calls | avg us | share % |", "|---|---|---|---|"]
            lines += [f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} |" for r in rows_pf]
    (dst / f"{tag}_summary.md").write_text("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
