"""DESIGN.md §0's current-state table from the round's evidence files (profiles/<tag>_bench_n1.jsonl, _per_shape.txt,
_pmc_table.txt, pmc_dominant_h2.json):   python scripts/design_state_table.py r06   -> markdown on stdout."""
import collections
import json
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
lines = [json.loads(l) for l in open("profiles/%s_bench_n1.jsonl" % tag) if l.startswith("{")]
head = lines[-1]
extra = {l["extra_workload"].split()[0]: l["line"] for l in lines if "extra_workload" in l}
pmc = json.load(open("profiles/pmc_dominant_h2.json"))
fam = collections.defaultdict(float)
for l in open("profiles/%s_per_shape.txt" % tag):
    m = re.match(r"(\S.*?)\s+grid=.*ms/step=\s*([\d.]+)", l)
    if not m:
        continue
    n, ms = m.group(1), float(m.group(2))
    a = re.findall(r"\d+", n[n.find("<"):]) if "<" in n else []
    if n.startswith("conv_h2"):
        key = "convs k=11" if a[0] == "11" else ("polyphase ConvTranspose" if a[6] == "2" else "other convs")
    elif "resblock" in n or "pair" in n:
        key = "fused ResBlock pairs"
    else:
        key = "everything else"
    fam[key] += ms
tot = sum(fam.values())
busy = {}
for l in open("profiles/%s_pmc_table.txt" % tag):
    p = l.split()
    if len(p) > 9 and re.match(r"(conv1d_h2|resblock_pair_h2)_kernel<", p[0]) and p[5] != "-":
        busy[p[0]] = (float(p[5]), float(p[9]), int(p[1]), float(p[4]))


def wavg(pred):
    w = [(b * n * c, n * c) for k, (b, m, n, c) in busy.items() if pred(k)]
    return sum(x for x, _ in w) / max(sum(y for _, y in w), 1e-9)


r = head["roofline"]


def row(*c):
    print("| " + " | ".join(c) + " |")


row("line", "value", "of its ceiling", "limiter")
row("---", "---", "---", "---")
row("**headline** configs[1], VITS B = 32", "**%.1f ms/step, %.3ge8 samples/s, %.0f× RT**" % (head["ms_per_step"], head["value"] / 1e8, head["rtf_x"]),
    "—", "sum of its conv launches")
row("dominant `conv1d_h2<11,1,1,4,4,1,0>`", "%.0f µs / launch, %.0f TF-eq" % (r["avg_launch_us"], r["achieved"]),
    "**%.3f** of 833 TF-eq" % r["frac"], "busy %.3f at %.2f GHz" % (pmc["mfma_busy_frac"], pmc["kernel_cycles"] / r["avg_launch_us"] / 1e3))
row("— its HBM traffic", "%.2f GB / launch" % (pmc["hbm_bytes_per_launch"] / 1e9),
    "%.2f× algorithmic; %.2f of 8 TB/s" % (pmc["hbm_bytes_per_launch"] / r["algorithmic_bytes_per_launch"], r["frac_of_8TBps"]), "halo, weight stream")
preds = {"fused ResBlock pairs": lambda s: s.startswith("resblock"), "convs k=11": lambda s: s.startswith("conv1d_h2_kernel<11"),
         "other convs": lambda s: s.startswith("conv1d_h2") and not s.startswith("conv1d_h2_kernel<11") and not s.endswith(",2>"),
         "polyphase ConvTranspose": lambda s: s.startswith("conv1d_h2") and s.endswith(",2>"), "everything else": lambda s: False}
lim = {"fused ResBlock pairs": "issue slots; + memory waits at k = 3, C = 32", "convs k=11": "issue + power cap", "other convs": "issue slots",
       "polyphase ConvTranspose": "store epilogue", "everything else": "HBM / launch latency"}
for k in preds:
    b = wavg(preds[k])
    row("share: %s" % k, "%.1f ms (%.0f %%)" % (fam[k], 100 * fam[k] / tot), ("busy %.2f" % b) if b else "—", lim[k])
c0, c2, b1, c4, ms = extra.get("configs[0]"), extra.get("configs[2]"), extra.get("vits_b1"), extra.get("configs[4]"), extra.get("mas")
if c0:
    row("configs[0], one sentence", "%.2f ms (%.2f, two in flight)" % (c0["ms_per_step"], c0["observed"]["two_lanes_ms_per_sentence"]),
        "%d launches" % c0["roofline"]["launches_per_request"], "launch chain")
if c2:
    rr = c2["roofline"]
    row("configs[2], HiFiGAN-v1 256 × 8192", "%.2f s/step, %.3ge8 samples/s" % (c2["ms_per_step"] / 1e3, c2["value"] / 1e8),
        "%.3f of 833; HBM set %.2f" % (rr["frac"], rr["hbm_subset_mean_frac_of_8TBps"]), "as the headline")
    row("— its HBM traffic", "%.1f kB / sample" % (rr["traffic_bytes_per_sample"] / 1e3),
        "%.2f× algorithmic (%.1f kB)" % (rr["traffic_excess"]["measured_over_algorithmic"], rr["algorithmic_bytes_per_sample"] / 1e3), "§5.6")
if b1:
    row("VITS B = 1 request", "**%.2f ms p50** (%.2f, two in flight)" % (b1["value"], b1["observed"]["two_lanes_ms_per_request"]),
        "%d launches" % b1["observed"]["launches_per_request"], "§5.4")
if c4:
    row("configs[4] vocoder half, first chunk", "%.2f ms" % c4["value"], "—", "launch chain")
if ms:
    row("MAS `[32,257,770]` (row a1)", "**%.3f ms**, %.3ge10 cells/s" % (ms["ms_per_step"], ms["value"] / 1e10),
        "%.3f of 8 TB/s at 12 B / cell" % ms["roofline"]["frac"], "serial chain, §5.5")
cb = head["cpu_baseline"]
row("CPU oracle beside the headline", "%.3ge5 samples/s, %d of %d cores" % (cb["value"] / 1e5, cb["cores"], cb.get("host_cores", 0)),
    "GPU / CPU %.0f×" % (head["value"] / cb["value"]), "—")
