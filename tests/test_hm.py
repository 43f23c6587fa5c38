"""The reference's two modified HM-16.15 codecs, compiled UNCHANGED against this repository's boundary
(tools/hm/Makefile: include/tf_compat in place of TensorFlow 1.9 + CPython 2.7, libpnn_hip.so in place of their libraries).

CPU tests: the build recipe works where the reference checkout exists, the four binaries need neither TensorFlow nor
libpython, the CPython shadow header reads a pickled float the way `loading.load_via_pickle` does, and an encoder started
without a GPU fails as loudly as the reference does when a graph cannot be loaded (no CPU fallback).
GPU tests (BASELINE.json configs[3] / [4] in small): encode one seeded synthetic 4:0:0 picture with seeded random-init
models, decode the bitstream, and require the decoder's picture to equal the encoder's reconstruction bit for bit --
in-process and with several concurrent encoders behind the batching service.
"""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HM = os.path.join(ROOT, "tools", "hm")
sys.path.insert(0, HM)
import run_hm  # noqa: E402

VARIANTS = ("substitution", "switch")
HAVE_REF = os.path.isdir("/root/reference/hevc/hm_16_15_substitution/source")


def _binaries():
    return [os.path.join(HM, "_build", v, "TApp%sStatic" % k) for v in VARIANTS for k in ("Encoder", "Decoder")]


@pytest.fixture(scope="module")
def hm_built():
    if HAVE_REF:
        subprocess.check_call(["make", "-s", "-C", HM, "-j8"])
    missing = [b for b in _binaries() if not os.path.exists(b)]
    if missing:
        pytest.skip("HM binaries not built (tools/hm/Makefile needs the reference checkout): %s" % missing[0])
    return True


def test_hm_links_without_tensorflow_or_python(hm_built):
    for b in _binaries():
        needed = subprocess.check_output(["readelf", "-d", b]).decode()
        libs = [l.split("[")[1].split("]")[0] for l in needed.splitlines() if "(NEEDED)" in l]
        assert "libpnn_hip.so" in libs
        assert not [l for l in libs if "tensorflow" in l or "python" in l or "protobuf" in l or "nsync" in l], libs
        syms = subprocess.check_output(["nm", "-D", "--undefined-only", b]).decode()
        assert " Py" not in syms and "tensorflow" not in syms        # the CPython / TF names are all header-only look-alikes
        for name in ("pnn_create_empty", "pnn_load_model_file", "pnn_predict_fc", "pnn_predict_conv", "pnn_client_predict_f32"):
            assert name in syms, "%s does not bind %s" % (os.path.basename(b), name)
        out = subprocess.run([b, "--help"], capture_output=True, text=True)
        assert "HM software" in out.stdout


def test_hm_options_of_the_reference_are_kept(hm_built):
    """The three command-line options the reference added (TAppEncCfg.cpp:1096-1098, TAppDecCfg.cpp:99-101)."""
    for b in _binaries():
        out = subprocess.run([b, "--help"], capture_output=True, text=True).stdout
        for opt in ("PathToAdditionalDirectory", "PathToMeanTraining", "PathToFilePathsToGraphsOutput"):
            assert opt in out


def test_python_shadow_reads_the_mean_like_pickle(tmp_path):
    """include/tf_compat/python2.7/Python.h through the call sequence HM makes: every pickle protocol of a float, the
    reference's own byte string, a text file; and the error paths (missing module / attribute / file, not a float)."""
    exe = str(tmp_path / "py_shadow")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-I" + os.path.join(ROOT, "include", "tf_compat"),
                           os.path.join(ROOT, "tests", "py_shadow_sample.cpp"), "-o", exe])
    mean = 117.8952234192841
    cases = {}
    for proto in range(0, pickle.HIGHEST_PROTOCOL + 1):
        p = tmp_path / ("mean_p%d.pkl" % proto)
        p.write_bytes(pickle.dumps(mean, protocol=proto))
        cases[str(p)] = mean
    ref_bytes = tmp_path / "mean_ref.pkl"
    ref_bytes.write_bytes(b"\x80\x02G@]yKW+\x1c\x10.")              # sets/results/training_set/means/luminance/mean_training.pkl
    cases[str(ref_bytes)] = mean
    txt = tmp_path / "mean.txt"
    txt.write_text("117.8952234192841\n")
    cases[str(txt)] = mean
    neg = tmp_path / "neg.pkl"
    neg.write_bytes(pickle.dumps(-1.0, protocol=2))                 # the value PyFloat_AsDouble also uses as its error code
    cases[str(neg)] = -1.0
    for path, want in cases.items():
        r = subprocess.run([exe, "loading", "load_via_pickle", path], capture_output=True, text=True)
        assert r.returncode == 0, (path, r.stderr)
        assert float(r.stdout) == want, (path, r.stdout)
    bad = tmp_path / "list.pkl"
    bad.write_bytes(pickle.dumps([1.0, 2.0], protocol=2))
    for argv, msg in ((["nosuch", "load_via_pickle", str(txt)], "ImportError"), (["loading", "nosuch", str(txt)], "AttributeError"),
                      (["loading", "load_via_pickle", str(tmp_path / "absent.pkl")], "IOError"),
                      (["loading", "load_via_pickle", str(bad)], "UnpicklingError")):
        r = subprocess.run([exe] + argv, capture_output=True, text=True)
        assert r.returncode == 1 and msg in r.stderr, (argv, r.returncode, r.stderr)


REF_TEST_EXE = os.path.join(HM, "_build", "hm_common", "executable")


def _reference_tree(tmp_path, models=()):
    """A scratch tree laid out like the reference checkout, as far as its test program reads it (it is run from the root
    with relative paths, hevc/hm_common/c++/README.md:18-33): the three small data files of hevc/hm_common/c++/pseudo_data
    (committed copies: tests/golden/hm_common_pseudo_data) and, in place of its frozen test graphs -- which are not in the
    checkout -- seeded .pnnw models under the very names it opens."""
    import shutil
    from context_adaptive_neural_network_based_prediction_amd import weights as wts
    root = tmp_path / "ref_root"
    data = root / "hevc" / "hm_common" / "c++" / "pseudo_data"
    (root / "hevc" / "hm_common" / "c++" / "pseudo_visualization").mkdir(parents=True)
    data.mkdir(parents=True)
    for f in os.listdir(os.path.join(ROOT, "tests", "golden", "hm_common_pseudo_data")):
        shutil.copy(os.path.join(ROOT, "tests", "golden", "hm_common_pseudo_data", f), str(data / f))
    params = {}
    for w, is_fc in models:
        (data / ("width_target_%d" % w)).mkdir()
        params[w] = wts.init_params(w, is_fc, seed=40 + w, bias_std=0.05)
        wts.save_pnnw(str(data / ("width_target_%d" % w) / "graph_output.pbtxt"), params[w], w, is_fc)
    return str(root), params


def _run_ref_test(root, *argv):
    r = subprocess.run([REF_TEST_EXE] + list(argv), cwd=root, capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout, r.stderr


def test_reference_test_program_on_the_boundary_cpu(hm_built, tmp_path):
    """The reference's OWN test program of its HM glue (hevc/hm_common/c++/source_test, built unchanged by tools/hm): the
    tests that need no network run here -- tensor creation through the look-alike, the context gather against the
    expectations printed beside it, the model-table parsers on the reference's fixtures, the embedded-Python calls
    (sys.path, get_callable, the pickled integer -1 whose value is also PyInt_AsLong's error code)."""
    if not os.path.exists(REF_TEST_EXE):
        pytest.skip("hm_common test program not built")
    root, _ = _reference_tree(tmp_path)
    rc, out, err = _run_ref_test(root, "create_tensors_flattened_context")
    assert rc == 0 and "Dimension of index 1: 80" in out and "Dimension of index 1: 320" in out
    rc, out, err = _run_ref_test(root, "create_tensors_context_portion")
    assert rc == 0 and out.count("Dimension of index 3: 1") == 6 and "Dimension of index 2: 192" in out and "Dimension of index 1: 128" in out
    rc, out, err = _run_ref_test(root, "parse_file_strings_three_keys")
    assert rc == 0
    for line in ("Key: {4, 0}, value: path_0", "Key: {32, 1}, value: path_4", "Key: {64, 2}, value: path_3", "Key: {8, 2}, value: path_2",
                 "Key: {32, 1}, value: path_1"):
        assert line in out, out
    rc, out, err = _run_ref_test(root, "parse_file_strings_one_key")
    assert rc == 0 and "Key: 0, value: path_1" in out and "Key: 1, value: path_0" in out
    rc, out, err = _run_ref_test(root, "append_sys_path", "hevc/hm_common")
    assert rc == 0 and "hevc/hm_common" in out
    rc, out, err = _run_ref_test(root, "get_callable", "hevc/hm_common")
    assert rc == 0 and "No error occurs" in out
    rc, out, err = _run_ref_test(root, "load_via_pickle", "hevc/hm_common")
    assert rc == 0 and "Loaded integer: -1" in out
    rc, out, err = _run_ref_test(root, "extract_context_portions")
    assert rc == 0 and out.count("Expected buffer:") >= 9
    rc, out, err = _run_ref_test(root, "check_overlapping_intra_pattern_context_portions")
    assert rc == 0 and "No error occurs" in out
    # without a GPU, loading a graph fails the way a missing TensorFlow graph does: an error Status, test returns -1
    import torch
    if not torch.cuda.is_available():
        root2, _ = _reference_tree(tmp_path / "b", models=((4, True),))
        rc, out, err = _run_ref_test(root2, "load_graph")
        assert rc != 0 and "no CPU fallback" in err


def _parse_block(out, w):
    rows = [l.split() for l in out.splitlines() if len(l.split()) == w]
    vals = []
    for r in rows:
        try:
            vals.append([float(v) for v in r])
        except ValueError:
            pass
    assert len(vals) >= w, out
    return np.array(vals[-w:], np.float32)


@pytest.mark.gpu
def test_reference_test_program_on_the_boundary_gpu(hm_built, tmp_path, oracle):
    """The network tests of the same program on the MI355X: load_graph / load_graphs, and its two prediction tests -- an FC
    4x4 net fed a flat -100 context with a 0 line in the column above the block, a conv 16x16 net fed -80 with a +20 line
    (tests.cpp:943-1156).  The trained graphs it was written for are not in the checkout, so the models are seeded
    random-init ones and the printed predictions are compared with the oracle on the same inputs instead of with the
    "second column close to 0 / 20" remark."""
    if not os.path.exists(REF_TEST_EXE):
        pytest.skip("hm_common test program not built")
    root, params = _reference_tree(tmp_path, models=((4, True), (8, True), (16, False)))
    for name in ("load_graph", "load_graphs"):
        rc, out, err = _run_ref_test(root, name)
        assert rc == 0, err
    rc, out, err = _run_ref_test(root, "prediction_neural_network_fully_connected")
    assert rc == 0, err
    w = 4
    ctx = np.full((1, 5 * w * w), -100.0, np.float32)
    for i in range(w):
        ctx[0, w + 1 + i * 3 * w] = 0.0
    want = oracle.fc_forward(params[4], w, ctx)[0]
    np.testing.assert_allclose(_parse_block(out, w), want, rtol=0, atol=2e-3)
    rc, out, err = _run_ref_test(root, "prediction_neural_network_convolutional")
    assert rc == 0, err
    w = 16
    above = np.full((1, w, 3 * w), -80.0, np.float32)
    left = np.full((1, 2 * w, w), -80.0, np.float32)
    above[0, :, w + 1] = 20.0
    want = oracle.conv_forward(params[16], w, above, left)[0]
    np.testing.assert_allclose(_parse_block(out, w), want, rtol=0, atol=2e-3)


def test_regular_hm_roundtrip_on_the_cpu(hm_built, tmp_path):
    """The harness itself, on the codec that needs no GPU: the reference's hm_16_15_regular (stock HM-16.15) encodes and
    decodes a synthetic 4:0:0 picture; decoder == encoder reconstruction, HM's picture hash agrees."""
    if not os.path.exists(os.path.join(HM, "_build", "regular", "TAppEncoderStatic")):
        pytest.skip("regular variant not built")
    res = run_hm.encode_decode("regular", run_hm.make_frame(128, 192, 5), 32, None, None, str(tmp_path))
    assert res["decoder_equals_encoder"] and not res["decoder_hash_error"] and res["psnr_rec_db"] > 30.0, res
    assert res["enc_pnn"] == {} and res["enc_total_time_s"] is not None


def test_hm_without_gpu_fails_like_a_missing_graph(hm_built, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    table, mean_path = run_hm.make_models(str(tmp_path / "models"), only_widths=(4,))
    with pytest.raises(RuntimeError) as e:
        run_hm.encode_decode("substitution", run_hm.make_frame(64, 64, 0), 32, table, mean_path, str(tmp_path))
    assert "no CPU fallback" in str(e.value) and "Assertion" in str(e.value)


def _check(res, variant):
    assert res["decoder_equals_encoder"], res
    assert not res["decoder_hash_error"], res
    assert res["psnr_rec_db"] > 25.0, res
    enc, dec = res["enc_pnn"], res["dec_pnn"]
    assert set(enc) == {4, 8, 16, 32, 64}, enc                      # five sessions, as TComPrediction.cpp:126-129
    assert all(enc[w]["runs"] > 0 for w in (4, 8, 16, 32, 64)), enc  # the fast search tries the PNN mode at every CU size
    assert sum(v["runs"] for v in dec.values()) > 0, "the encoder never selected the PNN mode: %r" % (res,)
    assert sum(v["runs"] for v in dec.values()) < sum(v["runs"] for v in enc.values())


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_hm_encode_decode_roundtrip_on_gpu(hm_built, variant, tmp_path):
    """configs[3] in small: the unmodified HM encoder of the reference, PNN on the MI355X through the TensorFlow
    look-alike, one 256x192 4:0:0 picture at QP 32 (intra_main_rext settings); the decoder must rebuild the encoder's
    reconstruction exactly (every PNN block it decodes was predicted by the same kernels in the same order)."""
    table, mean_path = run_hm.make_models(str(tmp_path / "models"))
    res = run_hm.encode_decode(variant, run_hm.make_frame(192, 256, 3), 32, table, mean_path, str(tmp_path))
    print(res)
    _check(res, variant)
    assert sum(v["cache_hits"] for v in res["enc_pnn"].values()) > 0   # the RD search repeats identical calls


@pytest.mark.gpu
def test_hm_switch_420_chroma_blocks(hm_built, tmp_path):
    """4:2:0 input through hm_16_15_switch: mode 35 is also a chroma candidate there (TComDataCU.cpp:1373-1375), so the
    luma-trained nets predict chroma blocks of width 4-32 whose contexts HM gathers with unitWidth = unitHeight = 2
    (TEncSearch.cpp:1197-1200).  All three planes of the decoder's picture must equal the encoder's."""
    table, mean_path = run_hm.make_models(str(tmp_path / "models"))
    y = run_hm.make_frame(128, 192, 31)
    cb = run_hm.make_frame(64, 96, 32)
    cr = run_hm.make_frame(64, 96, 33)
    res = run_hm.encode_decode("switch", y, 32, table, mean_path, str(tmp_path), chroma=(cb, cr))
    print(res)
    assert res["chroma_format"] == "420"
    assert res["decoder_equals_encoder"] and not res["decoder_hash_error"], res
    assert sum(v["runs"] for v in res["dec_pnn"].values()) > 0


@pytest.mark.gpu
def test_hm_with_the_trained_checkpoints(hm_built, tmp_path):
    """The only trained weights the reference ships (convolutional 4x4 / 8x8, tests/golden/conv{4,8}_single.pnnw) inside the
    codec: the model table points widths 4 and 8 at them (the look-alike accepts HM's flattened feed for a convolutional
    model, see pnn_tf_compat.h), widths 16-64 stay random-init.  A trained predictor is a real competitor of the 34 HEVC
    modes, so the encoder selects it on natural-looking content and the decoder must reproduce every such block."""
    table, mean_path = run_hm.make_models(str(tmp_path / "models"), trained_small=True)
    from tests import util
    frame = np.clip(util.make_plane(128, 192, seed=12).astype(np.int32), 0, 255).astype(np.uint8)   # smooth 8x8 patches + noise
    res = run_hm.encode_decode("substitution", frame, 27, table, mean_path, str(tmp_path))
    print(res)
    assert res["decoder_equals_encoder"] and not res["decoder_hash_error"], res
    assert res["enc_pnn"][4]["kind"].startswith("convolutional") and res["enc_pnn"][8]["kind"].startswith("convolutional")
    assert res["dec_pnn"][4]["runs"] + res["dec_pnn"][8]["runs"] > 0, "the trained 4x4 / 8x8 predictors were never selected: %r" % (res,)


@pytest.mark.gpu
def test_hm_concurrent_encodes_through_the_batching_service(hm_built, tmp_path):
    """configs[4] in small: four encoder processes share the GPU through ONE batching-service process
    (PNN_SERVICE_SOCKET; their single-block requests are coalesced into batched launches); each bitstream is then decoded
    by a stand-alone decoder that owns its own context -- the pictures must still match bit for bit, i.e. a block's
    prediction does not depend on the batch it travelled in (canonical_order)."""
    from concurrent.futures import ThreadPoolExecutor
    table, mean_path = run_hm.make_models(str(tmp_path / "models"))
    sock = str(tmp_path / "pnn.sock")
    srv = subprocess.Popen([sys.executable, "-m", "context_adaptive_neural_network_based_prediction_amd.service", "--socket", sock,
                            "--table", table, "--max-batch", "64"], cwd=ROOT, stdout=subprocess.PIPE, text=True)
    try:
        assert "listening" in srv.stdout.readline()
        frames = [run_hm.make_frame(128, 192, 20 + j) for j in range(4)]

        def job(j):
            return run_hm.encode_decode("switch", frames[j], 32, table, mean_path, str(tmp_path), tag=str(j),
                                        env={"PNN_SERVICE_SOCKET": sock}, decoder_env={})
        with ThreadPoolExecutor(4) as ex:
            results = list(ex.map(job, range(4)))
    finally:
        srv.terminate()
        tail = srv.stdout.read()
        srv.wait(20)
    print(tail)
    for r in results:
        print(r)
        assert r["decoder_equals_encoder"] and not r["decoder_hash_error"], r
        assert all("via service" in v["kind"] for v in r["enc_pnn"].values())
        assert all("via service" not in v["kind"] for v in r["dec_pnn"].values())
    import re
    m = re.search(r"(\d+) requests in (\d+) batched calls \(largest batch (\d+)\), (\d+) clients", tail)
    assert m, tail
    requests, calls, largest, clients = map(int, m.groups())
    assert clients == 20 and requests > 0                           # 4 encoders x 5 sessions
    assert calls < requests and largest >= 2                         # requests of concurrent encoders were coalesced


@pytest.mark.gpu
def test_hm_decoder_refuses_an_encoder_on_another_arithmetic(hm_built, tmp_path):
    """The arithmetic contract, enforced: an encoder behind a SPLIT-mode batching service; the service's tag
    (pnn_client_arithmetic_tag) travels to the decoder as $PNN_EXPECT_TAG.  A stand-alone decoder on the library's default float32
    refuses to start -- a clean "arithmetic mismatch" instead of a picture that drifts --, the same decoder switched to the split
    mode starts and reproduces the encoder's reconstruction bit for bit."""
    from context_adaptive_neural_network_based_prediction_amd import service
    table, mean_path = run_hm.make_models(str(tmp_path / "models"))
    sock = str(tmp_path / "pnn.sock")
    srv = subprocess.Popen([sys.executable, "-m", "context_adaptive_neural_network_based_prediction_amd.service", "--socket", sock,
                            "--table", table, "--max-batch", "64"], cwd=ROOT, stdout=subprocess.PIPE, text=True, env=dict(os.environ, PNN_PRECISION="1"))
    try:
        assert "listening" in srv.stdout.readline()
        cl = service.Client(sock)
        tags = {w: cl.arithmetic_tag(w) for w in (4, 8, 16, 32, 64)}
        cl.close()
        assert len(set(tags.values())) == 1 and "split-f16x3" in tags[4], tags
        frame = run_hm.make_frame(128, 192, 77)
        with pytest.raises(run_hm.ArithmeticMismatch) as err:                       # decoder on float32 (the default): refused
            run_hm.encode_decode("switch", frame, 32, table, mean_path, str(tmp_path / "a"), env={"PNN_SERVICE_SOCKET": sock}, decoder_env={},
                                 expect_tag=tags[4])
        assert "split-f16x3" in str(err.value) and ":f32:" in str(err.value), str(err.value)
        r = run_hm.encode_decode("switch", frame, 32, table, mean_path, str(tmp_path / "b"), env={"PNN_SERVICE_SOCKET": sock},
                                 decoder_env={"PNN_PRECISION": "1"}, expect_tag=tags[4])   # the same decoder on the encoder's arithmetic
        assert r["decoder_equals_encoder"] and not r["decoder_hash_error"], r
        assert all("via service" not in v["kind"] for v in r["dec_pnn"].values())
    finally:
        srv.terminate()
        srv.stdout.read()
        srv.wait(20)


@pytest.mark.gpu
def test_hm_encodes_dealt_over_one_service_per_device(hm_built, tmp_path):
    """The multi-GPU form of the HM path is replicas: one batching service per device, encoder j talks to service
    j % n_devices, no exchange between devices.  One GPU here, so both services sit on device 0 -- what is checked is the
    dealing: both services got clients, every picture decodes to the encoder's reconstruction."""
    import json, re
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hm", "run_hm.py"), "--variant", "switch", "--width", "192",
                          "--height", "128", "--jobs", "4", "--service", "--devices", "0,0", "--out", str(tmp_path / "run")],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    runs = [json.loads(l) for l in lines if l.startswith("{") and "decoder_equals_encoder" in l]
    assert len(runs) == 4 and all(r["decoder_equals_encoder"] and not r["decoder_hash_error"] for r in runs), out.stdout
    served = [int(m.group(1)) for m in re.finditer(r"(\d+) requests in \d+ batched calls \(largest batch \d+\), (\d+) clients", out.stdout)]
    clients = [int(m.group(2)) for m in re.finditer(r"(\d+) requests in \d+ batched calls \(largest batch \d+\), (\d+) clients", out.stdout)]
    assert len(served) == 2 and min(served) > 0, out.stdout
    assert clients == [20, 20], clients                              # 2 encoders + 2 decoders per service, 5 sessions each


@pytest.mark.parametrize("variant", VARIANTS)
def test_hm_with_the_pnn_on_host_cores_needs_no_gpu(hm_built, oracle, tmp_path, variant):
    """The reference's route in small, on a box without a GPU (BASELINE.json configs[0]: the CPU-runnable plumbing case): the unchanged HM
    binaries talk to the batching service through the client stub of libpnn_hip.so, and the service's backend answers from the CPU oracle
    (tools/hm/cpu_pnn_service.py -- bench.py's cpu_baseline leg of the HM campaigns).  One 64 x 64 picture: encode, decode, decoder ==
    encoder; the service answers seeded contexts of every width like the oracle called directly (Pel epilogue, socket framing, FC and
    conv request shapes)."""
    import campaign
    from tests import util
    name = "tiny_" + variant
    campaign.CONFIGS[name] = {"variant": variant, "pictures": 1, "width": 64, "height": 64, "baseline": "one 64 x 64 picture, PNN on host cores"}
    try:
        r = campaign.run_campaign(name, str(tmp_path / "work"), [0], backend="cpu", yardstick=False, spot_check=True, timeout=600, cpu_threads=2)
    finally:
        del campaign.CONFIGS[name]
    assert r["pnn_backend"] == "cpu" and r["pictures"] == 1 and r["every_decode_equals_its_encoder"] is True
    assert r["service"]["requests"] > 50
    for (_, w), rec in r["_spot_check"].items():
        flat, is_fc = run_hm.model_params(w)
        a, l = rec["above"], rec["left"]
        want = oracle.epilogue(oracle.fc_forward(flat, w, util.flatten_fc(a, l)) if is_fc else oracle.conv_forward(flat, w, a, l), run_hm.MEAN)
        assert np.array_equal(np.stack(rec["pel"]), want), "width %d" % w


@pytest.mark.gpu
@pytest.mark.parametrize("config,picture_set,arithmetic", [("kodak", "synthetic", "f32"), ("kodak", "natural", "f32"), ("bsds", "synthetic", "f32"), ("bsds", "natural", "f32"),
                                                           ("kodak", "natural", "split"), ("bsds", "synthetic", "split")])
def test_hm_campaign_at_stated_counts(hm_built, oracle, tmp_path, config, picture_set, arithmetic):
    """BASELINE.json configs[3] / configs[4] AS STATED: 24 pictures of 768 x 512 through hm_16_15_substitution, 100 pictures of
    480 x 320 through hm_16_15_switch (all widths 4-64), every encode in flight behind one batching service per device -- on the
    engineered synthetic pictures and on windows of the natural fixtures (widths 4 / 8 then run the reference's trained checkpoints).
    Every decoded picture must equal its encoder's reconstruction, and the service that served the campaign must answer seeded
    contexts of every width like the oracle (within 1 LSB: .5 ties) -- asked through the same socket before it stops.
    All four on the reference's arithmetic (float32, TComPrediction.cpp:572-579,601-608: the library's default since round 5), two of
    them on the split-f16 mode as well."""
    import campaign
    from tests import util
    if picture_set == "natural" and not os.path.exists(os.path.join(ROOT, "tests", "golden", "natural_luma.npz")):
        pytest.skip("tests/golden/natural_luma.npz is generated from the reference checkout by __graft_entry__.build()")
    r = campaign.run_campaign(config, str(tmp_path / "work"), [0], picture_set=picture_set, yardstick=False, spot_check=True, timeout=600, arithmetic=arithmetic)
    assert r["arithmetic"] == arithmetic
    assert r["pictures"] == campaign.CONFIGS[config]["pictures"] == (24 if config == "kodak" else 100)
    assert r["picture_size"] == ("768x512 4:0:0" if config == "kodak" else "480x320 4:0:0")
    assert r["variant"] == ("hm_16_15_substitution" if config == "kodak" else "hm_16_15_switch")
    assert r["every_decode_equals_its_encoder"] is True
    tags = r["arithmetic_tags"]                                      # asked through the service's socket; every decoder ran with it as $PNN_EXPECT_TAG
    assert tags["encoder_side"] == tags["decoder_side"] and len(tags["encoder_side"]) == 1 and tags["decoders_checked_expect_tag"] is True
    assert (":f32:" if arithmetic == "f32" else ":split-f16x3:") in tags["encoder_side"][0], tags
    assert r["service"]["requests"] > 10000 and r["service"]["backend_calls"] <= r["service"]["requests"]
    served = {w: v["session_run_calls"] for w, v in r["pnn_calls"]["enc_pnn"].items()}
    assert all(served[w] > 0 for w in (4, 8, 16, 32)), served        # every TU width reached the PNN (64: only in the switch variant's fast search)
    spot = r["_spot_check"]
    assert sorted(w for (_, w) in spot) == [4, 8, 16, 32, 64]
    for (_, w), rec in spot.items():
        flat, is_fc = run_hm.model_params(w, trained_small=(picture_set == "natural"))
        assert is_fc == rec["is_fc"]
        a, l = rec["above"], rec["left"]
        want = oracle.fc_forward(flat, w, util.flatten_fc(a, l)) if is_fc else oracle.conv_forward(flat, w, a, l)
        want = oracle.epilogue(want, run_hm.MEAN)
        got = np.stack(rec["pel"])
        assert np.abs(got.astype(np.int64) - want).max() <= 1, "width %d: the campaign's service differs from the oracle" % w
        assert (got != want).mean() < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["kodak", "bsds"])
def test_bench_hm_campaign_in_small(hm_built, config):
    """BASELINE.json configs[3] / configs[4] through bench.py (`--workload hm_kodak` / `hm_bsds`, here with 4 pictures instead of
    24 / 100): one batching service per device, all encodes in flight, hm_16_15_regular beside it; ONE JSON line (< 4 KB) whose `hm`
    record says that every decoded picture equals its encoder's reconstruction, everything else in the detail file.  The Kodak case
    also runs the cpu_baseline leg: the first two pictures with the PNN answered on host cores (CPU oracle behind the same service)
    beside the same two on the GPU."""
    import json
    detail = os.path.join(ROOT, "gpurun_out", "test_hm_detail_%s.json" % config)
    os.makedirs(os.path.dirname(detail), exist_ok=True)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "hm_" + config, "--hm-quick", "--detail-file", detail]
    if config == "bsds":
        cmd.append("--no-cpu-baseline")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    h = d["hm"]
    assert h["pictures"] == 4 and h["every_decode_equals_its_encoder"] is True
    assert h["variant"] == ("hm_16_15_substitution" if config == "kodak" else "hm_16_15_switch")
    full = json.load(open(detail))["hm"][config]
    assert full["service"]["requests"] > 1000 and full["service"]["backend_calls"] <= full["service"]["requests"]
    assert d["value"] == pytest.approx(full["service"]["pnn_blocks_per_s_over_the_wall"], rel=1e-5)
    assert sum(v["session_run_calls"] for v in full["pnn_calls"]["enc_pnn"].values()) > full["service"]["requests"]     # the rest were cache hits
    assert full["yardstick_hm_16_15_regular"]["every_decode_equals_its_encoder"] is True
    if config == "kodak":
        cb = d["cpu_baseline"]
        assert cb["unit"] == "pictures/s" and cb["kind"] == "port" and cb["value"] > 0 and cb["gpu_same_sample"] > 0
        cp = full["cpu_pnn"]
        assert cp["cpu"]["every_decode_equals_its_encoder"] is True and cp["gpu_same_sample"]["every_decode_equals_its_encoder"] is True
        assert cp["cpu"]["service"]["requests"] == cp["gpu_same_sample"]["service"]["requests"] or not cp["same_bits"]
    else:
        assert d["cpu_baseline"]["value"] is None
