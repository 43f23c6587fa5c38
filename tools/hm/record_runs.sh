#!/bin/bash
# Records the HM runs quoted in DESIGN.md section 5b into one text file (default profiles/r02_hm_runs.txt).  GPU box.
#   tools/hm/record_runs.sh [out-file]
out=${1:-profiles/r03_hm_runs.txt}
work=${TMPDIR:-/tmp}/hm_runs.$$
mkdir -p "$work"
run() {   # title, then run_hm.py arguments; result lines cut after the first fields
    echo "== $1" >> "$out"; shift
    python3 tools/hm/run_hm.py --out "$work/r" "$@" 2>&1 | grep -v amdgpu.ids | grep -v '^$' | cut -c1-900 >> "$out"
    rm -rf "$work/r"
}
cat > "$out" <<HDR
# HM-16.15 (reference sources, unchanged) on libpnn_hip.so: tools/hm/record_runs.sh on one MI355X box ($(nproc) host cores).
# Synthetic 4:0:0 pictures (run_hm.make_frame), seeded random-init models in the reference's architectures, intra_main_rext settings, QP 32.
# enc_total_time_s = HM's own 'Total Time' (clock(): CPU time of the encoder process -- time blocked on the service socket is not in it);
# enc_wall_s = wall clock of the encoder process; decoder_equals_encoder = decoded picture == encoder reconstruction, byte for byte;
# enc_pnn / dec_pnn = Session::Run calls per width and how many were answered from the prediction cache.
# Lines are cut after the first fields; "jobs ... wall_s" = all encodes + decodes of the run, "pnn service" = the server's own count.

HDR
# one throw-away encode first: the very first HIP process on a fresh box pays for the driver's own warm-up (~0.5 s)
python3 tools/hm/run_hm.py --out "$work/warm" --variant switch --width 192 --height 128 --jobs 1 > /dev/null 2>&1; rm -rf "$work/warm"
run "yardstick: hm_16_15_regular (the reference's stock HM-16.15, no PNN, CPU only), ONE synthetic 768x512 4:0:0 picture, QP 32" --variant regular --width 768 --height 512 --jobs 1
run "configs[3]-like: hm_16_15_substitution, ONE synthetic 768x512 4:0:0 picture, QP 32, in-process contexts" --variant substitution --width 768 --height 512 --jobs 1
run "the same through hm_16_15_switch" --variant switch --width 768 --height 512 --jobs 1
run "configs[3]-like: hm_16_15_substitution, 4 synthetic 768x512 pictures at a time, in-process contexts (4 x 5 on one GPU)" --variant substitution --width 768 --height 512 --jobs 4
run "the same 4 through ONE batching service process (encoders and decoders)" --variant substitution --width 768 --height 512 --jobs 4 --service
run "4:2:0 through hm_16_15_switch (the PNN also predicts chroma blocks, contexts in 2-pixel availability units), ONE 480x320 picture" --variant switch --width 480 --height 320 --jobs 1 --yuv420
run "with the reference's trained convolutional 4x4 / 8x8 checkpoints in the table (the only trained weights it ships), hm_16_15_switch, ONE 480x320 picture" --variant switch --width 480 --height 320 --jobs 1 --trained-small
for j in 1 4 16 32; do
    run "configs[4]-like: hm_16_15_switch, $j x 480x320 at a time, ONE batching service" --variant switch --width 480 --height 320 --jobs $j --service
done
run "configs[4]-like, the multi-GPU form (replicas): 16 x 480x320, TWO batching services (here both on device 0), encodes dealt round-robin" --variant switch --width 480 --height 320 --jobs 16 --service --devices 0,0
rm -rf "$work"
grep -c decoder_equals_encoder "$out"; grep -c '"decoder_equals_encoder": true' "$out"
