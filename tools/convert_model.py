#!/usr/bin/env python3
"""Convert the reference's trained models to the `.pnnw` format of libpnn_hip.so and rewrite a model table.

    python tools/convert_model.py --table hevc/hm_common/paths_to_graphs_output/single.txt --out models/ [--root <reference checkout>]

Every path of the `width,is_pair,channel,path` table (hevc/hm_common/c++/source_common/tools.cpp:52-111) may be a frozen
GraphDef (`graph_output.pbtxt`, binary), a TF V2 checkpoint prefix or a `.pnnw`; widths 4/8 are fully-connected and
16/32/64 convolutional, as TComPrediction.cpp:567-614 assumes.  Writes `<out>/<table name>` pointing at the new files.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_adaptive_neural_network_based_prediction_amd import weights as wts  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--root", default=".")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    entries = []
    for line in open(a.table):
        if not line.strip():
            continue
        f = [x for x in line.replace(";", ",").split(",") if x != ""]
        width, pair, ch, path = int(f[0]), int(f[1]), int(f[2]), f[3].strip()
        src = path if os.path.isabs(path) else os.path.join(a.root, path)
        name = "w%d_%s_ch%d.pnnw" % (width, "pair" if pair else "single", ch)
        wts.convert_model(src, os.path.join(a.out, name), width, width <= 8)
        entries.append((width, pair, ch, name))
        print("converted %s -> %s" % (src, name))
    print("table:", wts.write_model_table(os.path.join(a.out, os.path.basename(a.table)), entries))


if __name__ == "__main__":
    main()
