"""Text summary of the campaign lines tools/hm/record_campaigns.sh wrote (profiles/rNN_hm_runs.txt is its output):
    python tools/hm/summarize_campaigns.py gpurun_out/r05 > profiles/r05_hm_runs.txt"""
import json
import os
import sys

d = sys.argv[1]
print("# BASELINE configs[3] / configs[4] at their stated picture counts through the reference's two modified HM-16.15 codecs (compiled unchanged on")
print("# libpnn_hip.so), tools/hm/record_campaigns.sh on ONE MI355X box: every encode in flight behind one batching service, hm_16_15_regular beside it.")
print("# All eight campaigns of this file ran back to back on the same box; `x regular` = wall / wall of hm_16_15_regular on the same pictures.")
for cfg in ("kodak", "bsds"):
    for pics in ("synthetic", "natural"):
        rows = {}
        for ar in ("f32", "split"):
            p = os.path.join(d, "hm_%s_%s_%s_detail.json" % (cfg, pics, ar))
            if os.path.exists(p):
                rows[ar] = json.load(open(p))["hm"][cfg]
        if not rows:
            continue
        r0 = next(iter(rows.values()))
        print("\n== %s, %s pictures: %d pictures %s through %s" % (r0["config"].split(":")[0], pics, r0["pictures"], r0["picture_size"], r0["variant"]))
        for ar, r in rows.items():
            y = r.get("yardstick_hm_16_15_regular") or {}
            hc = r.get("host_cpu") or {}
            print("   %-5s wall %5.2f s all encodes + decodes (%5.2f pictures/s), %4.2f x hm_16_15_regular (%.2f s); every decode == its encoder: %s; bits %d (regular %s)"
                  % (ar, r["wall_s_all_encodes_and_decodes"], r["pictures_per_s"], r.get("wall_vs_regular") or 0, y.get("wall_s_all_encodes_and_decodes") or 0,
                     r["every_decode_equals_its_encoder"], r["bits_total"], y.get("bits_total")))
            sv = r["service"]
            pw = " ".join("%s: %.0f us x %.1f" % (w, 1e6 * v["backend_busy_s"] / max(v["calls"], 1), v["mean_batch"]) for w, v in sorted(sv["per_width"].items(), key=lambda kv: int(kv[0])))
            print("         service: %d requests in %d batched calls (mean batch %.2f, largest %d), %.0f PNN blocks/s over the wall; per width (time per call x mean batch): %s"
                  % (sv["requests"], sv["backend_calls"], sv["mean_batch"], sv["largest_batch"], sv["pnn_blocks_per_s_over_the_wall"], pw))
            print("         host: %d encodes in flight, service %.1f CPU-s (threads: %s), all processes %.1f CPU-s under a quota of %s CPUs, throttled %s times"
                  % (r["encodes_in_flight"], hc.get("service_cpu_s") or 0, hc.get("service_threads_cpu_s"), hc.get("all_processes_cpu_s") or 0, hc.get("cpu_quota"), hc.get("times_throttled")))
            cp = r.get("cpu_pnn")
            if cp:
                print("         PNN on host cores (CPU oracle behind the same service), first 2 pictures: %.1f s; the same 2 on the MI355X: %.2f s -> %.1f x; same bitstream sizes: %s"
                      % (cp["cpu"]["wall_s_all_encodes_and_decodes"], cp["gpu_same_sample"]["wall_s_all_encodes_and_decodes"], cp["gpu_over_cpu_wall"], cp["same_bits"]))
        if len(rows) == 2:
            a, b = rows["f32"], rows["split"]
            print("   f32 / split: wall x %.3f; bitstream sizes equal: %s" % (a["wall_s_all_encodes_and_decodes"] / b["wall_s_all_encodes_and_decodes"], a["bits_total"] == b["bits_total"]))
