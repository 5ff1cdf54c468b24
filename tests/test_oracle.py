"""CPU tests: the oracle (oracle/vit_seq_port.c) against golden vectors that the
REFERENCE ITSELF produced (oracle/make_golden.py ran the unmodified ViT_seq.c), and
-- when oracle/_ref is present -- against the reference live.  Bar: bit-exact."""
import numpy as np
import pytest

from pathlib import Path

from oracle import oracle as orc

GOLDEN = Path(__file__).resolve().parent / "golden"


def _check_summary(name, got, gold):
    got = np.ascontiguousarray(got, dtype=np.float32).ravel()
    stride = int(gold["stride"])
    assert got.size == int(gold[f"{name}_count"])
    assert np.array_equal(got[::stride], gold[f"{name}_sample"]), f"{name}: sampled values differ"
    assert got.astype(np.float64).sum() == float(gold[f"{name}_sum"]), f"{name}: fp64 sum differs"
    assert np.abs(got.astype(np.float64)).sum() == float(gold[f"{name}_abssum"])


def test_stages_bit_exact_vs_reference_goldens(oracle, weights, golden_stages):
    W, g = weights, golden_stages
    img = oracle.synth_image(0)
    conv = oracle.conv2d(img, W[1], W[2])
    _check_summary("conv", conv, g)
    tok = oracle.tokens_from_conv(conv, W[0], W[3])
    _check_summary("tokens", tok, g)
    ln = oracle.layer_norm(tok, W[4], W[5])
    _check_summary("ln", ln, g)
    _check_summary("mha", oracle.mha(ln, W[6], W[7], W[8], W[9]), g)
    _check_summary("mlp", oracle.mlp(ln, W[12], W[13], W[14], W[15]), g)
    _check_summary("enc0", oracle.encoder(tok, W[4:16]), g)
    head = oracle.linear(ln[:1], W[150], W[151], 1000)
    assert np.array_equal(head.ravel(), g["head"])
    assert np.array_equal(oracle.softmax(head.ravel()), g["softmax"])
    xs = (np.arange(4001, dtype=np.float32) - 2000) * np.float32(1 / 256)
    assert np.array_equal(oracle.gelu(xs), g["gelu"])


def test_attention_factoring_matches_mha(oracle, weights):
    """port_attention + the two projections == port_mha (which is pinned above)."""
    W = weights
    x = oracle.synth_fill(197 * 768, 77, 1.0, 0.0).reshape(197, 768)
    qkv = oracle.linear(x, W[6], W[7], 2304)
    out = oracle.linear(oracle.attention(qkv), W[8], W[9], 768)
    assert np.array_equal(out, oracle.mha(x, W[6], W[7], W[8], W[9]))


@pytest.mark.parametrize("index", [0, 3])
def test_full_model_bit_exact_vs_reference_goldens(oracle, weights, golden_full, index):
    """Whole forward for one image (~12 s each): logits and probabilities equal the
    reference's bit for bit."""
    logits, probs, _ = oracle.forward(oracle.synth_image(index), weights)
    assert np.array_equal(logits, golden_full["logits"][index])
    assert np.array_equal(probs, golden_full["probs"][index])
    assert abs(float(probs.sum()) - 1.0) < 1e-5


def test_real_image_bit_exact_vs_reference_golden(oracle, weights):
    """The reference's one real input (Data/input-1.bin: a normalised photograph, spatially correlated pixels with mean 1.2
    and std 0.5 -- unlike the iid synthetic images) through the port equals, bit for bit, what the reference's own
    ViT_seq.c gave for it on the same synthetic weights (tests/golden/b16_real_image.npz, oracle/make_golden.py real_image)."""
    g = np.load(GOLDEN / "b16_real_image.npz")
    assert g["image"].shape == (3, 224, 224) and g["image"].dtype == np.float32
    logits, probs, _ = oracle.forward(np.ascontiguousarray(g["image"]), weights)
    assert np.array_equal(logits, g["logits"][0]) and np.array_equal(probs, g["probs"][0])


def test_second_weight_set_bit_exact_vs_reference_golden(oracle):
    """A second draw of the synthetic weights (seed_base 1) and other images (100, 101): the port equals what the
    reference's own ViT_seq.c gave (tests/golden/b16_seed1.npz, oracle/make_golden.py other_seed) bit for bit -- the
    pinning does not rest on one weight set.  Image 101 only (12 s of CPU); the GPU test takes both."""
    g = np.load(GOLDEN / "b16_seed1.npz")
    w = oracle.synth_weights(int(g["seed_base"]))
    logits, probs, _ = oracle.forward(oracle.synth_image(int(g["first_image"]) + 1), w)
    assert np.array_equal(logits, g["logits"][1]) and np.array_equal(probs, g["probs"][1])


def test_answer_result_fixture_matches_goldens(golden_full):
    """tests/golden/b16_answer_result.txt is Main.c's output format (Main.c:71) and
    parses with comparator.c's sscanf pattern (comparator.c:15)."""
    import re
    from pathlib import Path
    lines = (Path(__file__).parent / "golden" / "b16_answer_result.txt").read_text().splitlines()
    assert len(lines) == golden_full["probs"].shape[0]
    for i, line in enumerate(lines):
        m = re.fullmatch(r"\[(\d+)\] label: (\d+) / prob: ([0-9.]+)", line)
        assert m and int(m.group(1)) == i
        assert int(m.group(2)) == int(golden_full["probs"][i].argmax())
        assert abs(float(m.group(3)) - float(golden_full["probs"][i].max())) < 1e-6


@pytest.mark.skipif(not orc.have_reference(), reason="oracle/_ref not built (needs /root/reference)")
def test_live_reference_stages(oracle, weights, tmp_path):
    """Re-run the reference's stage functions now and compare every element."""
    st = orc.run_reference("stages", 0, out_path=tmp_path / "stages.bin")
    W = weights
    img = oracle.synth_image(0)
    conv = oracle.conv2d(img, W[1], W[2])
    assert np.array_equal(conv.ravel(), st["conv"])
    tok = oracle.tokens_from_conv(conv, W[0], W[3])
    assert np.array_equal(tok.ravel(), st["tokens"])
    ln = oracle.layer_norm(tok, W[4], W[5])
    assert np.array_equal(ln.ravel(), st["ln"])
    assert np.array_equal(oracle.mha(ln, W[6], W[7], W[8], W[9]).ravel(), st["mha"])
    assert np.array_equal(oracle.mlp(ln, W[12], W[13], W[14], W[15]).ravel(), st["mlp"])
    assert np.array_equal(oracle.encoder(tok, W[4:16]).ravel(), st["enc0"])


def test_other_configs_run(oracle):
    """ViT-L/16 / H/14 shapes exist in the port ("parity unpinned": the reference has
    no code for them); only shape bookkeeping is checked here."""
    for name, tensors, toks in (("vit_l_16", 296, 197), ("vit_h_14", 392, 257)):
        o = orc.Oracle(name)
        assert o.num_tensors == tensors and o.tokens == toks
