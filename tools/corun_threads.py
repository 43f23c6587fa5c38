"""Builds seeded models in the reference's five architectures and runs tools/corun_threads.cpp on them (both arithmetics):
    python tools/corun_threads.py [seconds] [option=value ...]         (GPU box; tools/_bin/corun_threads is compiled here or travels)"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "hm"))
import run_hm
exe = os.path.join(ROOT, "tools", "_bin", "corun_threads")
libdir = os.path.join(ROOT, "context_adaptive_neural_network_based_prediction_amd")
if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(os.path.join(ROOT, "tools", "corun_threads.cpp")):
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "corun_threads.cpp"), "-o", exe,
                           "-L" + libdir, "-lpnn_hip", "-lpthread", "-Wl,-rpath," + libdir])
if "--build-only" in sys.argv:
    raise SystemExit(0)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
seconds = args[0] if args and "=" not in args[0] and args[0] not in ("fcprio", "queues") else "1.5"
opts = [a for a in args if "=" in a or a in ("fcprio", "queues")]
with tempfile.TemporaryDirectory() as d:
    table, _ = run_hm.make_models(os.path.join(d, "models"))
    for precision in ("0", "1"):
        sys.stdout.flush()
        subprocess.call([exe, table, precision, seconds] + opts)
