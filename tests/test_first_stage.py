"""N1 — first-stage decoders (qdiff/arch/first_stage.py) against outputs of the REAL reference Decoder
(tools/make_golden_first_stage.py: ldm/modules/diffusionmodules/model.py:465-572 with key-derived synthetic weights)."""
import pytest
import torch

from golden_util import load_fixture


def _build(case, cls_name):
    from qdiff import synthetic
    from qdiff.arch import first_stage as fs
    if cls_name == "kl":
        m = fs.AutoencoderKLDecoder(case["dd"], embed_dim=case["embed_dim"])
    else:
        m = fs.VQModelDecoder(case["dd"], embed_dim=case["embed_dim"], n_embed=64)
    sd = {k: synthetic.tensor_for(k, v.shape, seed=0) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    return m.eval()


@pytest.mark.parametrize("name,kind", [("kl_tiny", "kl"), ("vq_tiny", "vq")])
def test_decoder_is_bit_identical_to_the_reference_on_cpu(name, kind):
    """Same module tree (hence the same state-dict keys: the synthetic weights are derived from the KEYS), same operations
    in the same order: the CPU output equals the reference's bit for bit."""
    fx = load_fixture("first_stage.pt")
    case = fx[name]
    m = _build(case, kind)
    with torch.no_grad():
        out = m.decode(case["z"], force_not_quantize=True) if kind == "vq" else m.decode(case["z"])
    assert out.shape == case["out"].shape
    assert torch.equal(out, case["out"]), (out - case["out"]).abs().max().item()


def test_fused_attention_option_and_postprocessing():
    from qdiff.arch import first_stage as fs
    fx = load_fixture("first_stage.pt")
    case = fx["kl_tiny"]
    m = _build(case, "kl")
    for mod in m.modules():
        if isinstance(mod, fs.AttnBlock):
            mod.fused = True
    with torch.no_grad():
        out = m.decode(case["z"])
    assert (out - case["out"]).abs().max().item() <= 2e-5 * case["out"].abs().max().item()
    img = fs.decode_first_stage(_build(case, "kl"), case["z"] * 0.18215, scale_factor=0.18215, to_uint8=True)
    want = (torch.clamp((case["out"] + 1.0) / 2.0, 0.0, 1.0) * 255.0).round().to(torch.uint8)
    assert img.dtype == torch.uint8 and (img.int() - want.int()).abs().max().item() <= 1


def test_vector_quantiser_picks_the_nearest_codebook_entry():
    """taming's VectorQuantizer2 inference rule (restated, the dependency is not part of the reference tree): every
    latent vector is replaced by its nearest codebook entry; checked against brute-force squared distances."""
    from qdiff.arch import first_stage as fs
    g = torch.Generator().manual_seed(5)
    vq = fs.VectorQuantizer(128, 3)
    with torch.no_grad():
        vq.embedding.weight.copy_(torch.randn(128, 3, generator=g))
    z = torch.randn(2, 3, 5, 7, generator=g)
    zq = vq(z)
    assert zq.shape == z.shape
    rows = z.permute(0, 2, 3, 1).reshape(-1, 3)
    d = ((rows[:, None, :] - vq.embedding.weight[None]) ** 2).sum(-1)
    best = d.min(dim=1).values
    got = ((rows - zq.permute(0, 2, 3, 1).reshape(-1, 3)) ** 2).sum(-1)
    assert torch.allclose(got, best, rtol=1e-5, atol=1e-6)
    # decode() applies it unless told not to
    fx = load_fixture("first_stage.pt")
    m = _build(fx["vq_tiny"], "vq")
    with torch.no_grad():
        a = m.decode(fx["vq_tiny"]["z"])
        b = m.decode(m.quantize(fx["vq_tiny"]["z"]), force_not_quantize=True)
    assert torch.equal(a, b)


def test_first_stage_checkpoint_slice_loads():
    """`first_stage_model.*` keys of a full LDM / SD checkpoint (decoder.*, post_quant_conv.*, quantize.embedding.weight)
    load into the decode-side modules; encoder / loss keys are ignored; a missing decoder tensor is an error."""
    from qdiff.arch import first_stage as fs
    m, scale = fs.sd_v1_first_stage()
    assert scale == 0.18215
    sd = {"first_stage_model." + k: torch.zeros_like(v) for k, v in m.state_dict().items()}
    sd["first_stage_model.encoder.conv_in.weight"] = torch.zeros(1)
    sd["model.diffusion_model.out.2.weight"] = torch.zeros(1)
    m.load_first_stage_state_dict(sd)
    assert all(float(p.abs().sum()) == 0 for p in m.parameters())
    del sd["first_stage_model.decoder.conv_out.weight"]
    with pytest.raises(KeyError):
        m.load_first_stage_state_dict(sd)
    v, s = fs.lsun_beds_first_stage()
    assert s == 1.0 and v.quantize.embedding.weight.shape == (8192, 3)
    keys = set(m.state_dict())
    assert {"decoder.conv_in.weight", "decoder.mid.attn_1.q.weight", "decoder.up.3.upsample.conv.weight", "decoder.up.0.block.2.conv2.bias",
            "decoder.norm_out.weight", "post_quant_conv.weight"} <= keys


@pytest.mark.gpu
def test_decoder_on_gpu_matches_the_reference_golden(cuda):
    """channels-last MIOpen convolutions on the MI355X vs the reference's CPU output: fp32 accumulation-order noise only."""
    from qdiff.arch import first_stage as fs
    fx = load_fixture("first_stage.pt")
    for name, kind in (("kl_tiny", "kl"), ("vq_tiny", "vq")):
        case = fx[name]
        m = _build(case, kind).to(cuda)
        out = fs.decode_first_stage(m, case["z"].to(cuda), 1.0, force_not_quantize=True)
        assert (out.cpu() - case["out"]).abs().max().item() <= 1e-4 * case["out"].abs().max().item()


def test_decode_chunks_large_batches():
    """decode_first_stage splits the batch so that no activation exceeds max_activation_bytes; results are those of the
    unsplit call."""
    from qdiff.arch import first_stage as fs
    fx = load_fixture("first_stage.pt")
    case = fx["kl_tiny"]
    m = _build(case, "kl")
    z = torch.cat([case["z"], case["z"] * 0.5, -case["z"]], dim=0)           # 6 latents
    whole = fs.decode_first_stage(m, z, 1.0)
    per_image = fs.largest_activation_bytes(m.decoder, 8, 8)
    # ch = 32, ch_mult (1, 2, 2): the largest tensor is the nearest-2x copy in front of the last upsampling convolution,
    # 64 channels at 32 x 32 — twice the 32-channel output-resolution stream
    assert per_image == 64 * 32 * 32 * 4
    sd, _ = fs.sd_v1_first_stage()
    assert fs.largest_activation_bytes(sd.decoder, 64, 64) == 256 * 512 * 512 * 4      # 8 images = 2^31 bytes: chunks of 4
    beds, _ = fs.lsun_beds_first_stage()
    assert fs.largest_activation_bytes(beds.decoder, 64, 64) == 256 * 256 * 256 * 4    # 32 images = 2^31 bytes: chunks of 16
    parts = fs.decode_first_stage(m, z, 1.0, max_activation_bytes=2 * per_image)   # chunks of 2
    assert parts.shape == whole.shape and torch.equal(parts, whole)
