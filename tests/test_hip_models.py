"""Block- and model-level parity on a real MI355X against the GOLDEN VECTORS produced by the
reference's own modules (tests/golden/, oracle/gen_golden.py), in both precision configurations:

  fp32-class (fp32 activations, split-bf16 x3 MFMA)   tolerance: rel-L2 <= 1.6e-5 (blocks), 5e-5 (nets)
  bf16       (bf16 activations/weights, fp32 accum)    tolerance: rel-L2 <= 1.0e-2 (blocks), 2.5e-2 (nets)

Every bound is 1.5x the largest error MEASURED in its class on MI355X (round 2, CRG_TOL_REPORT=1 prints one line per
comparison: blocks fp32-class 1.04e-5 / bf16 6.6e-3, nets 3.0e-5 / 1.64e-2, SDXL nets bf16 2.17e-2, trajectories 6.0e-5):
a 2x regression fails.

and, for the VAE decoder, the north-star pixel bound: L-inf <= 1e-3 on clamp((x+1)/2, 0, 1).
"""
import os

import pytest
import torch

from cremage_amd.synth import synth_fill_, synth_input
from tests.conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda:0"
TOL_BLOCK = {torch.float32: 1.6e-5, BF: 1.0e-2}
TOL_NET = {torch.float32: 5e-5, BF: 2.5e-2}
TOL_SDXL_BF16 = 3.3e-2   # SDXL nets in bf16 (measured 2.17e-2 small / 1.74e-2 full size)
TOL_TRAJ = 1e-4          # fp32-class sampler trajectories, latents (measured <= 6.0e-5)


def prep(module, meta, dtype):
    synth_fill_(module, meta["seed"], prefix=meta["prefix"])
    return module.to(dtype).to(DEV).eval()


def img(x, dtype):
    return x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)


def close(got, ref, tol, what):
    got = got.detach().float().cpu()
    if got.dim() == 4:
        got = got.contiguous()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    r = rel_l2(got, ref)
    if os.environ.get("CRG_TOL_REPORT"):  # dev: one line per comparison, to set the tolerances from measurements
        print(f"\n[tol] {what}: rel-L2 {r:.3e} (bound {tol:.1e}, ratio {r / tol:.2f})")
    assert r < tol, (what, r, tol)
    return r


@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("tag", ["res_same", "res_skip", "res_320"])
def test_resblock(dtype, tag):
    from cremage_amd.ldm_hip.unet import ResBlock
    meta, g = load_golden("blk_" + tag)
    m = prep(ResBlock(meta["cin"], meta["emb"], 0, out_channels=meta["cout"]), meta, dtype)
    x = synth_input(tag + ".x", (2, meta["cin"], meta["hw"], meta["hw"]), meta["seed"])
    emb = synth_input(tag + ".emb", (2, meta["emb"]), meta["seed"])
    with torch.no_grad():
        y = m(img(x, dtype), emb.to(DEV).to(dtype))
    close(y, g["y"], TOL_BLOCK[dtype], tag)


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_resblock_concat_pair(dtype):
    """(h, skip) pair == cat([h, skip], dim=1) (openaimodel.py:808) through GN and the 1x1 skip conv"""
    from cremage_amd.ldm_hip.unet import ResBlock
    m = ResBlock(192, 256, 0, out_channels=64)
    synth_fill_(m, 7, prefix="pair.")
    m = m.to(dtype).to(DEV).eval()
    a, b = synth_input("pair.a", (2, 64, 6, 6), 7), synth_input("pair.b", (2, 128, 6, 6), 7)
    emb = synth_input("pair.emb", (2, 256), 7).to(DEV).to(dtype)
    with torch.no_grad():
        y1 = m((img(a, dtype), img(b, dtype)), emb)
        y2 = m(img(torch.cat([a, b], 1), dtype), emb)
    assert rel_l2(y1.float().cpu(), y2.float().cpu()) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_updown(dtype):
    from cremage_amd.ldm_hip.unet import Downsample, Upsample
    meta, g = load_golden("blk_downsample")
    m = prep(Downsample(64, True, out_channels=64), meta, dtype)
    with torch.no_grad():
        close(m(img(synth_input("down.x", (2, 64, 10, 10), meta["seed"]), dtype)), g["y"], TOL_BLOCK[dtype], "down")
    meta, g = load_golden("blk_upsample")
    m = prep(Upsample(64, True, out_channels=64), meta, dtype)
    with torch.no_grad():
        close(m(img(synth_input("up.x", (2, 64, 5, 5), meta["seed"]), dtype)), g["y"], TOL_BLOCK[dtype], "up")


@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("tag", ["ca_d40_m77", "ca_d80_m154", "ca_d160_self", "ca_d64_m77", "ca_d40_self"])
def test_cross_attention(dtype, tag):
    from cremage_amd.ldm_hip.transformer import CrossAttention
    meta, g = load_golden("op_" + tag)
    m = prep(CrossAttention(meta["query_dim"], meta["context_dim"], heads=meta["heads"], dim_head=meta["dim_head"]), meta, dtype)
    x = synth_input(tag + ".x", (2, meta["n"], meta["query_dim"]), meta["seed"]).to(DEV).to(dtype)
    ctx = synth_input(tag + ".ctx", (2, meta["m"], meta["context_dim"]), meta["seed"]).to(DEV).to(dtype) if meta["m"] else None
    with torch.no_grad():
        y = m(x, context=ctx)
        y_again = m(x, context=ctx)  # second call hits the K/V cache
    close(y, g["y"], TOL_BLOCK[dtype], tag)
    assert torch.equal(y, y_again)


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_cross_attention_lora_ipa(dtype):
    """LoRA folded into the packed weights + IP-Adapter FaceID second attention (attention.py:616-641,660-683)"""
    from cremage_amd.ldm_hip.transformer import CrossAttention
    meta, g = load_golden("op_ca_lora_ipa")
    tag = "ca_lora_ipa"
    m = CrossAttention(128, 96, heads=4, dim_head=32, lora_ranks=[4], lora_weights=[0.7], ipa_scale=0.6, ipa_num_tokens=4)
    synth_fill_(m, meta["seed"], prefix=meta["prefix"])
    with torch.no_grad():
        for name, p in m.named_parameters():
            if "_lora_" in name and p.ndim > 0:
                p.copy_(synth_input(tag + "." + name, p.shape, meta["seed"], 0.2))
    m = m.to(dtype).to(DEV).eval()
    x = synth_input(tag + ".x", (2, 50, 128), meta["seed"]).to(DEV).to(dtype)
    ctx = synth_input(tag + ".ctx", (2, 81, 96), meta["seed"]).to(DEV).to(dtype)
    with torch.no_grad():
        y = m(x, context=ctx)
    close(y, g["y"], TOL_BLOCK[dtype], tag)


def test_weight_updates_after_first_forward_are_seen():
    """The reference mutates a LIVE model: LoRA injection replaces parameters with `setattr(submodule, name, nn.Parameter(...))`
    (image_generator.py:408-453) and checkpoints arrive through `load_state_dict` (:345), both possibly after the model has
    already run.  Packed weights, LoRA merges and the fused Q|K weight are caches keyed on tensor identity + version, so the
    next forward must see the new values (and must equal a freshly built module holding them)."""
    import torch.nn as nn
    from cremage_amd.ldm_hip.transformer import CrossAttention
    mk = lambda: CrossAttention(128, None, heads=4, dim_head=32, lora_ranks=[4], lora_weights=[0.7])
    m = synth_fill_(mk(), 5, prefix="upd.").to(BF).to(DEV).eval()
    x = synth_input("upd.x", (2, 64, 128), 5).to(DEV).to(BF)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if "_lora_" in name and p.ndim > 0:
                p.copy_(synth_input("upd." + name, p.shape, 5, 0.2).to(p.dtype))
        y0 = m(x).float().cpu()
        # 1. LoRA injection the reference's way: a NEW Parameter object under the same name
        new_down = nn.Parameter(synth_input("upd.new_down", m.q_lora_downs[0].weight.shape, 6, 0.3).to(BF).to(DEV))
        setattr(m.q_lora_downs[0], "weight", new_down)
        y1 = m(x).float().cpu()
        assert rel_l2(y1, y0) > 1e-3
        # 2. a checkpoint loaded into the live module (in-place copy_: version bump, same identity)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        sd["to_v.weight"] = synth_input("upd.new_v", sd["to_v.weight"].shape, 7, 0.1).to(BF).to(DEV)
        sd["to_k.weight"] = synth_input("upd.new_k", sd["to_k.weight"].shape, 7, 0.1).to(BF).to(DEV)
        m.load_state_dict(sd)
        y2 = m(x).float().cpu()
        assert rel_l2(y2, y1) > 1e-3
        fresh = mk().to(BF).to(DEV).eval()
        fresh.load_state_dict(sd)
        y3 = fresh(x).float().cpu()
    assert torch.equal(y2, y3)


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_transformer_blocks(dtype):
    from cremage_amd.ldm_hip.transformer import BasicTransformerBlock, FeedForward, SpatialTransformer
    meta, g = load_golden("op_ff_geglu")
    m = prep(FeedForward(64, glu=True), meta, dtype)
    with torch.no_grad():
        close(m(synth_input("ff.x", (2, 30, 64), meta["seed"]).to(DEV).to(dtype)), g["y"], TOL_BLOCK[dtype], "ff")
    meta, g = load_golden("blk_basic_transformer")
    m = prep(BasicTransformerBlock(128, 4, 32, context_dim=96, checkpoint=False), meta, dtype)
    x = synth_input("btb.x", (2, 36, 128), meta["seed"]).to(DEV).to(dtype)
    ctx = synth_input("btb.ctx", (2, 77, 96), meta["seed"]).to(DEV).to(dtype)
    with torch.no_grad():
        close(m(x, context=ctx), g["y"], TOL_BLOCK[dtype], "btb")
    meta, g = load_golden("blk_spatial_transformer")
    m = prep(SpatialTransformer(128, 4, 32, depth=1, context_dim=96, use_checkpoint=False), meta, dtype)
    x = synth_input("st.x", (2, 128, 6, 6), meta["seed"])
    ctx = synth_input("st.ctx", (2, 77, 96), meta["seed"]).to(DEV).to(dtype)
    with torch.no_grad():
        close(m(img(x, dtype), context=ctx), g["y"], TOL_BLOCK[dtype], "st")


@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("name", ["unet_tiny", "unet_small_sd"])
def test_unet_small(dtype, name):
    from cremage_amd.ldm_hip.unet import UNetModel
    meta, g = load_golden(name)
    cfg = meta["cfg"]
    m = prep(UNetModel(**cfg), meta, dtype)
    x = synth_input(name + ".x", (meta["B"], 4, meta["L"], meta["L"]), meta["seed"]).to(DEV)
    ctx = synth_input(name + ".ctx", (meta["B"], meta["m"], cfg["context_dim"]), meta["seed"]).to(DEV)
    with torch.no_grad():
        y = m(x, timesteps=g["t"].to(DEV), context=ctx)
    assert y.dtype == torch.float32 and y.is_contiguous()
    close(y, g["y"], TOL_NET[dtype], name)


@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("shape", [(3, 24, 40), (1, 8, 72), (5, 40, 8)])
def test_unet_ragged_latents_vs_oracle(dtype, shape):
    """Non-square latents and odd batch sizes (portrait / landscape formats: every multiple of 8 is legal for the three
    stride-2 levels, openaimodel.py:780-816) - tile tails in M, partial XCD partitions, split-K tails - against the oracle run
    live on the same synthetic weights (the oracle itself is pinned on the square golden cases)."""
    from cremage_amd.ldm_hip.unet import UNetModel
    from oracle import ref_cpu as R
    from tests.conftest import synth_state_dict
    meta, _ = load_golden("unet_small_sd")
    cfg = meta["cfg"]
    B, H, W = shape
    m = UNetModel(**cfg)
    sd = synth_state_dict(m, meta["seed"], meta["prefix"])
    m = prep(m, meta, dtype)
    x = synth_input("ragged.x", (B, 4, H, W), meta["seed"])
    ctx = synth_input("ragged.ctx", (B, meta["m"], cfg["context_dim"]), meta["seed"])
    t = torch.tensor([981.0, 500.0, 37.0, 1.0, 999.0])[:B]
    with torch.no_grad():
        ref = R.unet_forward(sd, cfg, x, t, ctx)
        y = m(x.to(DEV), timesteps=t.to(DEV), context=ctx.to(DEV))
    close(y, ref, TOL_NET[dtype], f"unet ragged {shape}")


def test_vae_ragged_decode_encode_vs_oracle():
    """VAE decode / encode of a non-square latent, batch 3 (fp32-class path: split planes, asymmetric-pad stride-2 convs)."""
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    from oracle import ref_cpu as R
    from tests.conftest import synth_state_dict
    meta, _ = load_golden("vae_tiny")
    m = AutoencoderKL(meta["dd"], None, 4)
    sd = synth_state_dict(m, meta["seed"], meta["prefix"])
    m = prep(m, meta, torch.float32)
    z = synth_input("ragged.z", (3, 4, 10, 6), meta["seed"])
    im = synth_input("ragged.img", (3, 3, 12, 20), meta["seed"], 0.5).clamp(-1, 1)
    with torch.no_grad():
        close(m.decode(z.to(DEV)), R.autoencoder_decode(sd, meta["dd"], z), TOL_NET[torch.float32], "vae ragged dec")
        close(m.encode(im.to(DEV)).parameters, R.autoencoder_encode_moments(sd, meta["dd"], im), TOL_NET[torch.float32], "vae ragged enc")


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_unet_sd15_full(dtype):
    """Full-size SD1.5 UNet (859.52 M parameters), B=2, 64x64 latent - config 1's unit of work - against the
    output of the reference's own UNetModel on the same name-keyed synthetic weights."""
    from cremage_amd.ldm_hip.unet import UNetModel
    meta, g = load_golden("unet_sd15_full")
    cfg = meta["cfg"]
    m = UNetModel(**cfg)
    assert sum(p.numel() for p in m.parameters()) == meta["n_params"] == 859520964
    m = prep(m, meta, dtype)
    x = synth_input("unet_sd15_full.x", (2, 4, 64, 64), meta["seed"]).to(DEV)
    ctx = synth_input("unet_sd15_full.ctx", (2, 77, 768), meta["seed"]).to(DEV)
    with torch.no_grad():
        y = m(x, timesteps=g["t"].to(DEV), context=ctx)
    r = close(y, g["y"], TOL_NET[dtype], "unet_sd15_full")
    print(f"\n[parity] SD1.5 UNet full {dtype}: rel-L2 {r:.3e}, max-abs {(y.cpu() - g['y']).abs().max().item():.3e} "
          f"(|ref| max {g['y'].abs().max().item():.3f})")


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_vae_blocks(dtype):
    from cremage_amd.ldm_hip import vae as V
    meta, g = load_golden("blk_vae_resnet")
    m = prep(V.ResnetBlock(in_channels=64, out_channels=128, dropout=0.0, temb_channels=0), meta, dtype)
    with torch.no_grad():
        close(m(img(synth_input("vres.x", (2, 64, 8, 8), meta["seed"]), dtype)), g["y"], TOL_BLOCK[dtype], "vres")
    meta, g = load_golden("blk_vae_attn")
    m = prep(V.AttnBlock(64), meta, dtype)
    with torch.no_grad():
        close(m(img(synth_input("vattn.x", (2, 64, 6, 6), meta["seed"]), dtype)), g["y"], TOL_BLOCK[dtype], "vattn")
    meta, g = load_golden("blk_vae_downsample")
    m = prep(V.Downsample(64, True), meta, dtype)
    with torch.no_grad():
        close(m(img(synth_input("vdown.x", (2, 64, 10, 10), meta["seed"]), dtype)), g["y"], TOL_BLOCK[dtype], "vdown")
    meta, g = load_golden("blk_vae_upsample")
    m = prep(V.Upsample(64, True), meta, dtype)
    with torch.no_grad():
        close(m(img(synth_input("vup.x", (2, 64, 5, 5), meta["seed"]), dtype)), g["y"], TOL_BLOCK[dtype], "vup")


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_vae_tiny(dtype):
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    meta, g = load_golden("vae_tiny")
    m = prep(AutoencoderKL(meta["dd"], None, 4), meta, dtype)
    z = synth_input("vae_tiny.z", (2, 4, 8, 8), meta["seed"]).to(DEV)
    im = synth_input("vae_tiny.img", (2, 3, 16, 16), meta["seed"], 0.5).clamp(-1, 1).to(DEV)
    noise = synth_input("vae_tiny.noise", (2, 4, 8, 8), meta["seed"]).to(DEV)
    with torch.no_grad():
        close(m.decode(z), g["dec"], TOL_NET[dtype], "vae dec")
        post = m.encode(im)
        close(post.parameters, g["moments"], TOL_NET[dtype], "vae moments")
        close(post.sample(noise), g["sample"], TOL_NET[dtype], "vae sample")


def test_vae_sd15_full_decode_pixels():
    """North-star bound: VAE-decoded pixels within 1e-3 (L-inf, pixels in [0,1]) of the reference's fp32 CPU
    path, full-size SD1.5 decoder (49.49 M parameters), 64x64 latent -> 512x512 image."""
    from cremage_amd import ops
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    meta, g = load_golden("vae_sd15_full_decode")
    m = prep(AutoencoderKL(meta["dd"], None, 4), meta, torch.float32)
    z = synth_input("vae_full.z", (1, 4, 64, 64), meta["seed"]).to(DEV)
    with torch.no_grad():
        dec = m.decode(z / meta["scale_factor"])
        pix = ops.affine_cast(dec, 0.5, 0.5, torch.float32, 0.0, 1.0).cpu()
    dec = dec.cpu()
    assert dec.shape == (1, 3, 512, 512)
    sub = (dec[:, :, ::8, ::8] - g["dec_sub"]).abs().max().item()   # exact fp32 subsample of the reference output
    ref_pix = ((g["dec_f16"].float() + 1) / 2).clamp(0, 1)            # full image, stored as fp16 (<= 5e-4 rounding)
    linf = (pix - ref_pix).abs().max().item()
    print(f"\n[parity] SD1.5 VAE decode fp32-class: L-inf on the fp32 subsample {sub / 2:.3e} (pixel units), "
          f"full-image L-inf vs fp16-stored reference {linf:.3e}")
    assert sub / 2 < 1e-3
    assert linf < 1e-3 + 2.5e-4 * max(1.0, float(g["pix_stats"][2]))


def test_vae_sd15_full_decode_pixels_mx(monkeypatch):
    """The same full-size decode with the ResnetBlock convs on MX planes (CRG_PREC_F16MX, opt-in: ops.VAE_MX) - one fp16 pass plus fp8 cross
    terms per conv instead of three bf16 passes: the pixel bound 1e-3 must hold with margin (CPU emulation of the scheme: 8e-5)."""
    from cremage_amd import ops
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    monkeypatch.setattr(ops, "VAE_MX", True)
    meta, g = load_golden("vae_sd15_full_decode")
    m = prep(AutoencoderKL(meta["dd"], None, 4), meta, torch.float32)
    z = synth_input("vae_full.z", (1, 4, 64, 64), meta["seed"]).to(DEV)
    with torch.no_grad():
        dec = m.decode(z / meta["scale_factor"]).cpu()
    sub = (dec[:, :, ::8, ::8] - g["dec_sub"]).abs().max().item()
    print(f"\n[parity] SD1.5 VAE decode on MX planes: L-inf on the fp32 subsample {sub / 2:.3e} (pixel units)")
    assert sub / 2 < 2.5e-4


def test_vae_sd15_full_encode():
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    meta, g = load_golden("vae_sd15_full_encode")
    m = prep(AutoencoderKL(meta["dd"], None, 4), meta, torch.float32)
    x = synth_input("vae_full.img", (1, 3, 256, 256), meta["seed"], 0.5).clamp(-1, 1).to(DEV)
    with torch.no_grad():
        mom = m.encode(x).parameters
    close(mom, g["moments"], TOL_NET[torch.float32], "vae full encode")


@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("attn_type", ["vanilla", "vanilla-xformers"])
def test_sgm_vae_tiny(dtype, attn_type):
    """SDXL first stage against fixtures made by the reference's sgm Encoder / Decoder (sgm/modules/diffusionmodules/model.py:
    492-760; AttnBlock :161-195).  Both attn_type spellings (sd_xl_base.yaml:82 uses "vanilla-xformers") build the one HIP block."""
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    meta, g = load_golden("sgm_vae_tiny")
    m = prep(AutoencoderKL(dict(meta["dd"], attn_type=attn_type), None, 4), meta, dtype)
    z = synth_input("sgm_vae_tiny.z", (2, 4, 8, 8), meta["seed"]).to(DEV)
    im = synth_input("sgm_vae_tiny.img", (2, 3, 16, 16), meta["seed"], 0.5).clamp(-1, 1).to(DEV)
    noise = synth_input("sgm_vae_tiny.noise", (2, 4, 8, 8), meta["seed"]).to(DEV)
    with torch.no_grad():
        close(m.decode(z), g["dec"], TOL_NET[dtype], "sgm vae dec")
        post = m.encode(im)
        close(post.parameters, g["moments"], TOL_NET[dtype], "sgm vae moments")
        close(post.sample(noise), g["sample"], TOL_NET[dtype], "sgm vae sample")


def test_sgm_vae_full_decode_1024():
    """Full-size SDXL VAE: 128x128 latent -> 1024x1024 (C3 / C5's decode), fp32-class, against the reference's sgm Decoder run in
    fp32 on the CPU; pixel L-inf bound 1e-3 on the exact fp32 16x-subsample, and the whole every-8th-pixel grid vs its fp16 copy."""
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    meta, g = load_golden("sgm_vae_full")
    m = prep(AutoencoderKL(dict(meta["dd"], attn_type="vanilla-xformers"), None, 4), meta, torch.float32)
    z = synth_input("sgm_vae_full.z", (1, 4, 128, 128), meta["seed"]).to(DEV)
    with torch.no_grad():
        dec = m.decode(z / meta["scale_factor"]).cpu()
    assert dec.shape == (1, 3, 1024, 1024)
    e16 = (dec[:, :, ::16, ::16] - g["dec_sub16"]).abs().max().item() / 2
    e8 = (dec[:, :, ::8, ::8] - g["dec_sub8_f16"].float()).abs().max().item() / 2
    print(f"\n[parity] SDXL VAE decode 1024x1024 fp32-class: pixel L-inf {e16:.3e} on the exact subsample, {e8:.3e} on the fp16-stored grid")
    assert e16 < 1e-3
    assert e8 < 1e-3 + 2.5e-4 * max(1.0, float(g["dec_stats"][2]))
    x = synth_input("sgm_vae_full.img", (1, 3, 256, 256), meta["seed"], 0.5).clamp(-1, 1).to(DEV)
    with torch.no_grad():
        mom = m.encode(x).parameters
    close(mom, g["moments"], TOL_NET[torch.float32], "sgm vae full encode")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_c1_sd15_full_20_step_trajectory(mode):
    """BASELINE configs[0] at full size: SD1.5 512x512, batch 1, 20-step Euler, CFG 7.5 - the WHOLE trajectory (40 UNet
    evaluations through the k-diffusion stack + VAE decode) against the reference's fp32 CPU run (tests/golden/
    traj_c1_sd15_full.npz, ~5 min of CPU in the build container).  `fp32` = the fp32-class path end to end; `bf16` = the
    configuration bench.py times (bf16 UNet, fp32-class VAE): its ACCUMULATED error over 20 steps is what north_star's
    "stated fp32 per-pixel tolerance" is about, so it is measured, printed and bounded here."""
    import os
    from cremage_amd import pipeline as P
    from tests.conftest import GOLD
    if not os.path.exists(os.path.join(GOLD, "traj_c1_sd15_full.npz")):
        pytest.skip("full-size C1 fixture not generated")
    meta, g = load_golden("traj_c1_sd15_full")
    udt = torch.float32 if mode == "fp32" else BF
    ldm = P.build_synthetic_ldm(meta["unet"], meta["dd"], DEV, unet_dtype=udt, vae_dtype=torch.float32, seed=meta["seed"])
    B, L, seed = meta["B"], meta["L"], meta["seed"]
    c = synth_input("c1.c", (B, 77, 768), seed).to(DEV)
    uc = synth_input("c1.uc", (B, 77, 768), seed).to(DEV)
    x0 = synth_input("c1.x0", (B, 4, L, L), seed).to(DEV)
    sig = g["sigmas"].float()
    seen = {}
    orig = ldm.apply_model
    n = [0]

    def spy(x, t, cond):  # the latent entering UNet call i, un-scaled (apply_model gets x / sqrt(sigma^2 + 1))
        i = n[0]
        if i in (5, 10, 15):
            seen[i] = (x[:B].float() * float((sig[i] ** 2 + 1.0) ** 0.5)).cpu()
        n[0] += 1
        return orig(x, t, cond)

    ldm.apply_model = spy
    images, x = P.txt2img(ldm, c, uc, steps=meta["S"], sampler="euler", cfg_scale=meta["cfg"], height=8 * L, width=8 * L, x0=x0)
    assert n[0] == meta["S"]
    drift = {i: rel_l2(seen[i], g[f"x{i}"]) for i in (5, 10, 15)}
    lat = rel_l2(x.float().cpu(), g["x"])
    ref_pix = ((g["img_f16"].float() + 1) / 2).clamp(0, 1)
    pix = (images.cpu() - ref_pix).abs()
    sub = (images.cpu()[:, :, ::8, ::8] - ((g["img_sub8"] + 1) / 2).clamp(0, 1)).abs().max().item()
    print(f"\n[parity] C1 full-size 20-step Euler, {mode}: latent rel-L2 after 5/10/15/20 steps "
          f"{drift[5]:.3e} / {drift[10]:.3e} / {drift[15]:.3e} / {lat:.3e}; pixel L-inf {pix.max().item():.3e} "
          f"(exact fp32 subsample {sub:.3e}), pixel mean-abs {pix.mean().item():.3e}")
    if mode == "fp32":
        assert lat < 1.4e-4 and sub < 2.5e-4, (lat, sub)   # measured 9.0e-5 / 1.62e-4
    else:
        # measured: latent rel-L2 6.47e-2, pixel mean-abs 1.51e-2, pixel L-inf 0.133 (bf16 UNet x 20 steps; the VAE is fp32-class)
        assert lat < 9.7e-2 and pix.mean().item() < 2.3e-2 and pix.max().item() < 0.2, (lat, pix.mean().item(), pix.max().item())


def test_fp16_operand_build():
    """The fp16-operand build of the library (libcrg_hip_f16.so, CRG_HALF=f16: the same kernels with the _f16 matrix instructions,
    the dtype of the reference's own GPU flow, image_generator.py:489-493,748-751) in a child process: per-op errors at fp16
    round-off (three more mantissa bits than bf16: about 8x below the bf16 bounds of tests/test_hip_ops.py) and the full-size C1
    trajectory, whose end-to-end error is what VERDICT r2 asked to see next to the bf16 figure of
    test_c1_sd15_full_20_step_trajectory (bf16: latent rel-L2 6.5e-2, pixel L-inf 0.13, mean-abs 1.5e-2)."""
    import json
    import subprocess
    import sys
    from tests.conftest import REPO
    env = dict(os.environ, CRG_HALF="f16")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "_half_f16_run.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("HALF_F16_RESULT ")][-1]
    res = json.loads(line[len("HALF_F16_RESULT "):])
    print("\n[parity] fp16-operand build:", json.dumps(res))
    # bounds = 1.5x the errors measured on MI355X (round 3): ops 2.1e-4, attention 3.7e-4; C1 latent rel-L2 8.1e-3, pixel L-inf 1.5e-2,
    # pixel mean-abs 1.9e-3 - each about 8x below its bf16 counterpart, as three more mantissa bits predict
    for k in ("linear", "conv3x3", "groupnorm_pre", "ln_linear"):
        assert res[k] < 3.2e-4, (k, res[k])
    assert res["attention_4096_d40"] < 5.5e-4, res
    if "c1_latent_rel_l2" in res:
        assert res["c1_latent_rel_l2"] < 1.25e-2 and res["c1_pixel_linf"] < 2.3e-2 and res["c1_pixel_mean_abs"] < 2.9e-3, res


@pytest.mark.parametrize("nm", ["euler", "euler_a"])
def test_trajectory(nm):
    """5 sampler steps + decode through cremage_amd.pipeline (PyTorch sampler loop around the HIP UNet/VAE)
    against the trajectory the reference's own EulerSampler / EulerAncestralSampler stack produced."""
    from cremage_amd import pipeline as P
    meta, g = load_golden("traj_" + nm)
    ldm = P.build_synthetic_ldm(meta["unet"], meta["dd"], DEV, unet_dtype=torch.float32, vae_dtype=torch.float32, seed=meta["seed"])
    B, L, seed = meta["B"], meta["L"], meta["seed"]
    c = synth_input("traj.c", (B, 77, 96), seed).to(DEV)
    uc = synth_input("traj.uc", (B, 77, 96), seed).to(DEV)
    x0 = synth_input("traj.x0", (B, 4, L, L), seed).to(DEV)
    noises = iter([synth_input(f"traj.noise{i}", (B, 4, L, L), seed).to(DEV) for i in range(5)])
    images, x = P.txt2img(ldm, c, uc, steps=meta["S"], sampler=nm, cfg_scale=meta["cfg"], height=8 * L, width=8 * L, x0=x0,
                          noise_sampler=(lambda s, sn: next(noises)))
    close(x, g["x"], TOL_TRAJ, "traj latent " + nm)
    ref_img = ((g["img"] + 1) / 2).clamp(0, 1)
    assert (images.cpu() - ref_img).abs().max().item() < 2e-3


def test_trajectory_hires_latent():
    """Hires-fix with the latent upscaler (image_generator.py:958-999) vs the reference's sampler stack"""
    from cremage_amd import pipeline as P
    meta, g = load_golden("traj_hires_latent")
    ldm = P.build_synthetic_ldm(meta["unet"], meta["dd"], DEV, unet_dtype=torch.float32, vae_dtype=torch.float32, seed=meta["seed"])
    B, L, seed, f = meta["B"], meta["L"], meta["seed"], meta["factor"]
    c, uc = synth_input("hires.c", (B, 77, 96), seed).to(DEV), synth_input("hires.uc", (B, 77, 96), seed).to(DEV)
    x0 = synth_input("hires.x0", (B, 4, L, L), seed).to(DEV)
    noise = synth_input("hires.noise", (B, 4, f * L, f * L), seed).to(DEV)
    images, x, base = P.txt2img_hires(ldm, c, uc, steps=meta["S"], sampler="euler", cfg_scale=meta["cfg"], height=8 * L, width=8 * L,
                                      factor=f, strength=meta["strength"], x0=x0, fwd_noise=noise)
    close(base, g["base"], TOL_TRAJ, "hires base latent")
    close(x, g["x"], TOL_TRAJ, "hires latent")
    ref_img = ((g["img"] + 1) / 2).clamp(0, 1)
    assert (images.cpu() - ref_img).abs().max().item() < 3e-3


def test_trajectory_ddim_img2img():
    from cremage_amd import pipeline as P
    meta, g = load_golden("traj_ddim_img2img")
    ldm = P.build_synthetic_ldm(meta["unet"], meta["dd"], DEV, unet_dtype=torch.float32, vae_dtype=torch.float32, seed=meta["seed"])
    B, L, seed = meta["B"], meta["L"], meta["seed"]
    c = synth_input("traj.c", (B, 77, 96), seed).to(DEV)
    uc = synth_input("traj.uc", (B, 77, 96), seed).to(DEV)
    img_in = synth_input("traj.img", (B, 3, 32, 32), seed, 0.5).clamp(-1, 1).to(DEV)
    enc_noise = synth_input("traj.encnoise", (B, 4, L, L), seed).to(DEV)
    fwd_noise = synth_input("traj.fwdnoise", (B, 4, L, L), seed).to(DEV)
    # S=20, t_enc=3 <=> strength 0.15 (int(0.15*20) = 3, image_generator.py:727)
    images, x = P.img2img(ldm, img_in, c, uc, steps=meta["S"], strength=0.15, cfg_scale=meta["cfg"], enc_noise=enc_noise,
                          fwd_noise=fwd_noise)
    close(x, g["x"], TOL_TRAJ, "ddim latent")
    ref_img = ((g["img"] + 1) / 2).clamp(0, 1)
    assert (images.cpu() - ref_img).abs().max().item() < 2e-3


# ------------------------------------------------------------------------------------------ SDXL (sgm) twins
@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("name", ["sgm_unet_tiny", "sgm_unet_small_sdxl"])
def test_sgm_unet_small(dtype, name):
    """sgm UNetModel (label_emb, per-level transformer depth up to 10, linear proj_in/out, 64-wide heads) against the
    output of the reference's sgm UNetModel"""
    from cremage_amd.sgm_hip.unet import UNetModel
    meta, g = load_golden(name)
    cfg = meta["cfg"]
    m = prep(UNetModel(**cfg), meta, dtype)
    x = synth_input(name + ".x", (meta["B"], 4, meta["L"], meta["L"]), meta["seed"]).to(DEV)
    ctx = synth_input(name + ".ctx", (meta["B"], meta["m"], cfg["context_dim"]), meta["seed"]).to(DEV)
    y = synth_input(name + ".y", (meta["B"], cfg["adm_in_channels"]), meta["seed"]).to(DEV)
    with torch.no_grad():
        out = m(x, timesteps=g["t"].to(DEV), context=ctx, y=y)
    close(out, g["y"], TOL_SDXL_BF16 if dtype == BF else TOL_NET[dtype], name)


def test_sdxl_trajectory_euler_edm():
    """5 EulerEDMSampler steps (DiscreteDenoiser + EpsScaling + VanillaCFG) + VAE decode vs the reference's own stack"""
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    from cremage_amd.sgm_hip.sampling import DiffusionEngine
    from cremage_amd.sgm_hip.unet import UNetModel
    meta, g = load_golden("traj_sdxl_euler_edm")
    unet = synth_fill_(UNetModel(**meta["unet"]), meta["seed"], prefix="sgm_unet.")
    vae = synth_fill_(AutoencoderKL(meta["dd"], None, 4), meta["seed"], prefix="vae.")
    eng = DiffusionEngine(unet, vae, meta["scale_factor"]).to(DEV).eval()
    B, L, seed = meta["B"], meta["L"], meta["seed"]
    c = {"crossattn": synth_input("sgmtraj.c", (B, 77, 128), seed).to(DEV), "vector": synth_input("sgmtraj.cv", (B, 96), seed).to(DEV)}
    uc = {"crossattn": synth_input("sgmtraj.uc", (B, 77, 128), seed).to(DEV), "vector": synth_input("sgmtraj.ucv", (B, 96), seed).to(DEV)}
    x0 = synth_input("sgmtraj.x0", (B, 4, L, L), seed).to(DEV)
    x = eng.sample(x0, c, uc, meta["S"], meta["cfg"])
    close(x, g["x"], TOL_TRAJ, "sdxl traj latent")
    img = eng.decode_first_stage(x)
    assert (img.cpu() - g["img"]).abs().max().item() < 4e-3


def test_sdxl_img2img_trajectory():
    """SDXL img2img (BASELINE config 5's face-fix re-entry): VAE encode -> pruned-schedule Euler-EDM -> decode vs the reference"""
    from cremage_amd import pipeline as P
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    from cremage_amd.sgm_hip.sampling import DiffusionEngine
    from cremage_amd.sgm_hip.unet import UNetModel
    meta, g = load_golden("traj_sdxl_img2img")
    unet = synth_fill_(UNetModel(**meta["unet"]), meta["seed"], prefix="sgm_unet.")
    vae = synth_fill_(AutoencoderKL(meta["dd"], None, 4), meta["seed"], prefix="vae.")
    eng = DiffusionEngine(unet, vae, meta["scale_factor"]).to(DEV).eval()
    B, L, seed = meta["B"], meta["L"], meta["seed"]
    c = {"crossattn": synth_input("sgmi2i.c", (B, 77, 128), seed).to(DEV), "vector": synth_input("sgmi2i.cv", (B, 96), seed).to(DEV)}
    uc = {"crossattn": synth_input("sgmi2i.uc", (B, 77, 128), seed).to(DEV), "vector": synth_input("sgmi2i.ucv", (B, 96), seed).to(DEV)}
    img = synth_input("sgmi2i.img", (B, 3, 2 * L, 2 * L), seed, 0.5).clamp(-1, 1).to(DEV)
    en, fn = synth_input("sgmi2i.enc_noise", (B, 4, L, L), seed).to(DEV), synth_input("sgmi2i.noise", (B, 4, L, L), seed).to(DEV)
    close(eng.encode_first_stage(img, en), g["z"], TOL_NET[torch.float32], "sdxl img2img z")
    images, x = P.img2img_sdxl(eng, img, c, uc, steps=meta["S"], strength=meta["strength"], cfg_scale=meta["cfg"], enc_noise=en, fwd_noise=fn)
    close(x, g["x"], TOL_TRAJ, "sdxl img2img latent")
    ref = ((g["img"] + 1) / 2).clamp(0, 1)
    assert (images.cpu() - ref).abs().max().item() < 2e-3


def test_c5_chain_txt2img_then_facefix_reentry():
    """BASELINE config 5 as one chain (tiny): SDXL txt2img -> crop -> resize -> img2img re-entry (strength 0.3) -> paste, against the
    same chain assembled from the reference's own sampler / denoiser / guider / UNet / VAE pieces (oracle/gen_golden.py g_c5_chain)."""
    from cremage_amd import pipeline as P
    meta, g = load_golden("traj_c5_chain")
    eng = P.build_synthetic_sdxl(meta["unet"], meta["dd"], DEV, unet_dtype=torch.float32, vae_dtype=torch.float32, seed=meta["seed"])
    B, L, seed = meta["B"], meta["L"], meta["seed"]
    c = {"crossattn": synth_input("c5.c", (B, 77, 128), seed).to(DEV), "vector": synth_input("c5.cv", (B, 96), seed).to(DEV)}
    uc = {"crossattn": synth_input("c5.uc", (B, 77, 128), seed).to(DEV), "vector": synth_input("c5.ucv", (B, 96), seed).to(DEV)}
    x0 = synth_input("c5.x0", (B, 4, L, L), seed).to(DEV)
    en = synth_input("c5.enc_noise", (B, 4, L, L), seed).to(DEV)
    fn = synth_input("c5.noise", (B, 4, L, L), seed).to(DEV)
    boxes = [tuple(b) for b in meta["boxes"]]
    # the two passes use different step counts in the fixture (5 then 10 x 0.3): run them as the pipeline does, one call each
    first, x1 = P.txt2img_sdxl(eng, c, uc, steps=meta["S1"], cfg_scale=meta["cfg"], height=2 * L, width=2 * L, x0=x0)
    close(x1, g["x1"], TOL_TRAJ, "c5 first-pass latent")
    assert (first.cpu() - g["first"]).abs().max().item() < 2e-3
    import torch.nn.functional as F
    crops = torch.stack([F.interpolate(first[i:i + 1, :, t:t + sz, l:l + sz], size=(2 * L, 2 * L), mode="bilinear", align_corners=False)[0]
                         for i, (t, l, sz) in enumerate(boxes)])
    fixed, x2 = P.img2img_sdxl(eng, crops * 2.0 - 1.0, c, uc, steps=meta["S2"], strength=meta["strength"], cfg_scale=meta["cfg"],
                               enc_noise=en, fwd_noise=fn)
    close(x2, g["x2"], TOL_TRAJ, "c5 second-pass latent")
    assert (fixed.cpu() - g["fixed"]).abs().max().item() < 3e-3
    # and the one-call form (same step count for both passes, as the application runs it) is self-consistent with its parts
    out, f1, f2 = P.txt2img_sdxl_facefix(eng, c, uc, boxes, steps=meta["S2"], cfg_scale=meta["cfg"], height=2 * L, width=2 * L,
                                         fix_size=2 * L, strength=meta["strength"], x0=x0, enc_noise=en, fwd_noise=fn)
    assert out.shape == f1.shape == (B, 3, 2 * L, 2 * L) and torch.isfinite(out).all()
    for i, (t, l, sz) in enumerate(boxes):
        outside = torch.ones_like(out[i], dtype=torch.bool)
        outside[:, t:t + sz, l:l + sz] = False
        assert torch.equal(out[i][outside], f1[i][outside])          # untouched outside the box
        assert not torch.equal(out[i, :, t:t + sz, l:l + sz], f1[i, :, t:t + sz, l:l + sz])


@pytest.mark.parametrize("dtype", [BF])
def test_sgm_unet_sdxl_full(dtype):
    """Full-size SDXL UNet (2 567.46 M parameters), B=2, 128x128 latent (1024^2 image) vs the reference's sgm UNetModel"""
    import os
    from cremage_amd.sgm_hip.unet import UNetModel
    from tests.conftest import GOLD
    if not os.path.exists(os.path.join(GOLD, "sgm_unet_sdxl_full.npz")):
        pytest.skip("full-size SDXL fixture not generated")
    meta, g = load_golden("sgm_unet_sdxl_full")
    cfg = meta["cfg"]
    m = UNetModel(**cfg)
    assert sum(p.numel() for p in m.parameters()) == meta["n_params"] == 2567463684
    m = prep(m, meta, dtype)
    name = "sgm_unet_sdxl_full"
    x = synth_input(name + ".x", (2, 4, 128, 128), meta["seed"]).to(DEV)
    ctx = synth_input(name + ".ctx", (2, 77, 2048), meta["seed"]).to(DEV)
    y = synth_input(name + ".y", (2, 2816), meta["seed"]).to(DEV)
    with torch.no_grad():
        out = m(x, timesteps=g["t"].to(DEV), context=ctx, y=y)
    r = close(out, g["y"], TOL_SDXL_BF16, name)
    print(f"\n[parity] SDXL UNet full {dtype}: rel-L2 {r:.3e}, max-abs {(out.cpu() - g['y']).abs().max().item():.3e} "
          f"(|ref| max {g['y'].abs().max().item():.3f})")


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_controlnet_hook_attaches(dtype):
    """'ControlNet hooks still attach': a subclass written exactly like the reference's ControlledUnetModel
    (cldm.py:28-70 - it walks time_embed / input_blocks / middle_block / output_blocks / out itself, adds the control
    residuals in place and concatenates skips with th.cat) runs on the HIP module tree and matches the reference."""
    from cremage_amd.ldm_hip.nn import timestep_embedding
    from cremage_amd.ldm_hip.unet import UNetModel

    class ControlledUnetModel(UNetModel):
        def forward(self, x, timesteps=None, context=None, control=None, only_mid_control=False, **kwargs):
            hs = []
            with torch.no_grad():
                t_emb = timestep_embedding(timesteps, self.model_channels, repeat_only=False)
                emb = self.time_embed(t_emb)
                h = x.type(self.dtype)
                for module in self.input_blocks:
                    h = module(h, emb, context)
                    hs.append(h)
                h = self.middle_block(h, emb, context)
            if control is not None:
                h += control.pop()
            for i, module in enumerate(self.output_blocks):
                if only_mid_control or control is None:
                    h = torch.cat([h, hs.pop()], dim=1)
                else:
                    h = torch.cat([h, hs.pop() + control.pop()], dim=1)
                h = module(h, emb, context)
            h = h.type(x.dtype)
            return self.out(h)

    meta, g = load_golden("hook_controlnet")
    cfg = meta["cfg"]
    m = prep(ControlledUnetModel(**cfg), meta, dtype)
    m.dtype = dtype  # the reference's `use_fp16` switch (openaimodel.py:532) - here: the activation dtype of the walk
    B, L, seed = meta["B"], meta["L"], meta["seed"]
    x = synth_input("cn.x", (B, 4, L, L), seed).to(DEV).to(dtype)
    ctx = synth_input("cn.ctx", (B, 77, cfg["context_dim"]), seed).to(DEV).to(dtype)
    control = [synth_input(f"cn.control{i}", (B,) + tuple(s), seed, 0.3).to(DEV).to(dtype) for i, s in enumerate(meta["shapes"])]
    t = g["t"].to(DEV)
    with torch.no_grad():
        y = m(x, timesteps=t, context=ctx, control=[c.clone() for c in control])
        y_mid = m(x, timesteps=t, context=ctx, control=[c.clone() for c in control], only_mid_control=True)
    close(y, g["y"], TOL_NET[dtype], "controlnet hook")
    close(y_mid, g["y_mid"], TOL_NET[dtype], "controlnet hook (mid only)")


@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("name", ["controlnet_tiny", "controlnet_small_sd"])
def test_controlnet_fast_path(dtype, name):
    """the HIP ControlNet (cldm.py:73-342) and ControlLDM.apply_model (:374-393) against the reference's
    own modules: the control tensors, eps with unit / non-trivial control_scales, only_mid_control; and the hint cache."""
    from cremage_amd.cldm_hip import ControlLDM, ControlledUnetModel, ControlNet
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    meta, g = load_golden(name)
    cfg = meta["cfg"]
    cn = ControlNet(hint_channels=3, **{k: v for k, v in cfg.items() if k != "out_channels"})
    un = ControlledUnetModel(**cfg)
    synth_fill_(cn, meta["seed"], prefix=meta["cn_prefix"])
    synth_fill_(un, meta["seed"], prefix=meta["unet_prefix"])
    vae = AutoencoderKL(dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2], num_res_blocks=1,
                             attn_resolutions=[], dropout=0.0), None, 4)
    ldm = ControlLDM(cn, "hint", False, un, vae)
    ldm = ldm.to(dtype).to(DEV).eval()
    B, L, seed = meta["B"], meta["L"], meta["seed"]
    x = synth_input("cnet.x", (B, 4, L, L), seed).to(DEV)
    hint = (synth_input("cnet.hint", (B, 3, 8 * L, 8 * L), seed, 0.5).clamp(-1, 1) * 0.5 + 0.5).to(DEV)
    ctx = synth_input("cnet.ctx", (B, 77, cfg["context_dim"]), seed).to(DEV)
    t = g["t"].to(DEV)
    with torch.no_grad():
        control = ldm.control_model(x=x, hint=hint, timesteps=t, context=ctx)
        assert len(control) == meta["n_control"]
        for i, c in enumerate(control):
            close(c, g[f"control{i}"], TOL_NET[dtype], f"control{i}")
        cond = {"c_crossattn": [ctx], "c_concat": [hint]}
        close(ldm.apply_model(x, t, cond), g["eps"], TOL_NET[dtype], "eps")
        assert ldm.control_model._hint_cache is not None and ldm.control_model._hint_cache[0][0] is hint  # second call: cached
        ldm.control_scales = list(meta["scales"])
        close(ldm.apply_model(x, t, cond), g["eps_scaled"], TOL_NET[dtype], "eps (control_scales)")
        ldm.control_scales = [1.0] * meta["n_control"]
        ldm.only_mid_control = True
        close(ldm.apply_model(x, t, cond), g["eps_mid"], TOL_NET[dtype], "eps (only_mid_control)")
        # a changed hint image must not hit the cache
        hint.mul_(0.5)
        g2 = ldm.control_model.guided_hint(hint, control[0].dtype)
        hint.mul_(2.0)
        g1 = ldm.control_model.guided_hint(hint, control[0].dtype)
        assert rel_l2(g2.float().cpu(), g1.float().cpu()) > 1e-2


def test_controlnet_sd15_full():
    """Full-size ControlNet (cldm_v15.yaml: 361 M params) + SD1.5 UNet, bf16, B=2, L=64, 512x512 hint vs the reference."""
    from cremage_amd import pipeline as P
    meta, g = load_golden("controlnet_sd15_full")
    ldm = P.build_synthetic_control_ldm(device=DEV, seed=meta["seed"])
    assert sum(p.numel() for p in ldm.control_model.parameters()) == meta["n_params"]
    B, L, seed = meta["B"], meta["L"], meta["seed"]
    x = synth_input("cnet.x", (B, 4, L, L), seed).to(DEV)
    hint = (synth_input("cnet.hint", (B, 3, 8 * L, 8 * L), seed, 0.5).clamp(-1, 1) * 0.5 + 0.5).to(DEV)
    ctx = synth_input("cnet.ctx", (B, 77, 768), seed).to(DEV)
    t = g["t"].to(DEV)
    with torch.no_grad():
        control = ldm.control_model(x=x, hint=hint, timesteps=t, context=ctx)
        for i, c in enumerate(control):
            close(c[:, :, ::4, ::4], g[f"control{i}_sub"], TOL_NET[BF], f"control{i}")
        eps = ldm.apply_model(x, t, {"c_crossattn": [ctx], "c_concat": [hint]})
    close(eps, g["eps"], TOL_NET[BF], "eps")


def test_reference_gpu_calling_convention_half_autocast():
    """The reference's GPU flow (image_generator.py:489-493, :748-751): `model.half()` then `torch.autocast` around a sampler
    that feeds fp32 latents.  The drop-in must take fp16 PARAMETERS (packed to bf16 images), run its bf16 path, return x.dtype
    (openaimodel.py:810), and agree with the reference's fp32 output to the bf16 tolerance; same for the VAE decode."""
    from cremage_amd.ldm_hip.unet import UNetModel
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    meta, g = load_golden("unet_tiny")
    cfg = meta["cfg"]
    m = synth_fill_(UNetModel(**cfg), meta["seed"], prefix=meta["prefix"]).half().to(DEV).eval()
    assert next(m.parameters()).dtype == torch.float16
    x = synth_input("unet_tiny.x", (meta["B"], 4, meta["L"], meta["L"]), meta["seed"]).to(DEV)
    ctx = synth_input("unet_tiny.ctx", (meta["B"], meta["m"], cfg["context_dim"]), meta["seed"]).to(DEV)
    with torch.no_grad(), torch.autocast(device_type="cuda"):
        y = m(x, timesteps=g["t"].to(DEV), context=ctx)
    assert y.dtype == torch.float32
    close(y, g["y"], TOL_NET[BF], "unet half+autocast")
    vm, vg = load_golden("vae_tiny")
    vae = synth_fill_(AutoencoderKL(vm["dd"], None, 4), vm["seed"], prefix="vae.").half().to(DEV).eval()
    z = synth_input("vae_tiny.z", (2, 4, 8, 8), vm["seed"]).to(DEV)
    with torch.no_grad(), torch.autocast(device_type="cuda"):
        dec = vae.decode(z)
    assert dec.dtype == torch.float32 and torch.isfinite(dec).all()
    close(dec, vg["dec"], TOL_NET[BF], "vae half+autocast")
    # half parameters, fp32-class compute: only the fp16 rounding of the weights separates it from the golden
    vae.compute_dtype = torch.float32
    with torch.no_grad():
        dec32 = vae.decode(z)
    close(dec32, vg["dec"], 1.1e-3, "vae half params, fp32-class compute")  # measured 7.1e-4 (the PARAMETERS are fp16-rounded here)


def test_c4_unit_img2img_768_properties():
    """BASELINE.json configs[3]'s per-GPU unit at full size (SD1.5 img2img 768x768, 2 images, DDIM, strength 0.75 ->
    t_enc = 15 of 20 steps, VAE encode + decode) through size-independent properties: determinism (bitwise), and
    sample independence (an image computed in a batch of 2 == the same image computed alone) - the property the
    batch-sharded multi-GPU mode rests on."""
    from cremage_amd import pipeline as P
    ldm = P.build_synthetic_ldm(device=DEV, seed=99)
    b = 2
    img = (synth_input("c4.img", (b, 3, 768, 768), 44, 0.5).clamp(-1, 1)).to(DEV)
    c = synth_input("c4.c", (b, 77, 768), 7).to(DEV)
    uc = synth_input("c4.uc", (1, 77, 768), 7).expand(b, -1, -1).contiguous().to(DEV)
    en = synth_input("c4.en", (b, 4, 96, 96), 45).to(DEV)
    fn = synth_input("c4.fn", (b, 4, 96, 96), 46).to(DEV)
    run = lambda sl: P.img2img(ldm, img[sl], c[sl], uc[sl], steps=20, strength=0.75, cfg_scale=7.5, enc_noise=en[sl], fwd_noise=fn[sl])
    im_a, z_a = run(slice(0, 2))
    im_b, z_b = run(slice(0, 2))
    assert im_a.shape == (2, 3, 768, 768) and z_a.shape == (2, 4, 96, 96)
    assert torch.isfinite(im_a).all() and im_a.min() >= 0 and im_a.max() <= 1
    assert torch.equal(im_a, im_b) and torch.equal(z_a, z_b)
    im_1, z_1 = run(slice(1, 2))
    # bf16 UNet: split-K / tile shapes differ with the batch, so equality is to bf16 round-off, not bitwise
    assert rel_l2(z_1.cpu(), z_a[1:2].cpu()) < 3e-2
    assert (im_1.cpu() - im_a[1:2].cpu()).abs().mean().item() < 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_time_rows_hoisted_out_of_the_call(dtype):
    """`UNetModel.time_rows` computes the timestep-only work (sinusoid -> time_embed -> the ResBlocks' emb_layers, openaimodel.py:793-796,
    :222-228) for a whole schedule at once; a call whose timesteps carry their rows (ops.attach_time_rows, what the Euler samplers of
    this package do) must give the in-call result within the block tolerance, eagerly and through a hipGraph replay that receives
    DIFFERENT rows per replay; rows computed by another network (the ControlNet case) must be ignored."""
    from cremage_amd import ops
    from cremage_amd.graphs import GraphedModule
    from cremage_amd.ldm_hip.unet import UNetModel
    meta, g = load_golden("unet_small_sd")
    cfg = meta["cfg"]
    m = prep(UNetModel(**cfg), meta, dtype)
    x = synth_input("trow.x", (4, 4, 16, 16), 5).to(DEV)
    ctx = synth_input("trow.ctx", (4, 77, cfg["context_dim"]), 6).to(DEV)
    t_table = torch.tensor([[801.5] * 4, [333.25] * 4, [12.0] * 4], device=DEV)
    with torch.no_grad():
        rows = m.time_rows(t_table)
        assert rows.shape[:2] == (3, 4) and rows.dtype == torch.float32
        ref = [m(x.clone(), timesteps=t_table[i].clone(), context=ctx) for i in range(3)]
        got = [m(x.clone(), timesteps=ops.attach_time_rows(t_table[i].clone(), rows[i], m), context=ctx) for i in range(3)]
        # the rows replace the computation: garbage timesteps with the right rows give the right answer
        junk = m(x.clone(), timesteps=ops.attach_time_rows(torch.zeros(4, device=DEV), rows[1], m), context=ctx)
        other = m(x.clone(), timesteps=ops.attach_time_rows(t_table[1].clone(), torch.zeros_like(rows[1]), object()), context=ctx)
    tol = TOL_BLOCK[dtype]
    for i in range(3):
        close(got[i], ref[i].float().cpu(), tol, f"time rows, step {i}")
    assert torch.equal(junk, got[1]) and torch.equal(other, ref[1])
    assert not torch.allclose(got[0].float(), got[1].float())
    gm = GraphedModule(m, scratch_bytes=64 << 20)
    with torch.no_grad():
        yg = [gm(x.clone(), timesteps=ops.attach_time_rows(t_table[i].clone(), rows[i], m), context=ctx) for i in range(3)]
        yp = gm(x.clone(), timesteps=t_table[2].clone(), context=ctx)  # no rows: its own graph, the in-call embedding
    assert all(torch.equal(a, b) for a, b in zip(yg, got)) and torch.equal(yp, ref[2])
    assert gm.captures == 2 and gm.replays == 2


@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("model", ["unet", "controlled"])
def test_cfg_shared_prefix_matches_full_batch(dtype, model):
    """A batch-doubled call marked with ops.mark_cfg_dup (what the sampler wrapper does for classifier-free guidance,
    ldm_wrapper_for_k_diffusion.py:67-93) runs conv_in .. the first self-attention on ONE half: the result must be the full-batch
    result within the block tolerances (same arithmetic per sample; only tile configurations may differ with the batch size), for the
    plain UNet, the ControlNet-hooked subclass with control residuals, the `cfg_dup=True` keyword, and through a hipGraph replay;
    an unmarked call with DIFFERENT halves must be untouched by the feature."""
    from cremage_amd import ops
    from cremage_amd.graphs import GraphedModule
    from cremage_amd.ldm_hip.unet import UNetModel
    from cremage_amd.cldm_hip.cldm import ControlledUnetModel
    meta, g = load_golden("unet_small_sd")
    cfg = meta["cfg"]
    cls = UNetModel if model == "unet" else ControlledUnetModel
    m = prep(cls(**cfg), meta, dtype)
    assert m._cfg_split_index() == 1
    xh = synth_input("cfgdup.x", (2, 4, 16, 16), 5).to(DEV)
    x = torch.cat([xh, xh])
    t = torch.tensor([801.5, 333.25, 801.5, 333.25], device=DEV)
    ctx = synth_input("cfgdup.ctx", (4, 77, cfg["context_dim"]), 6).to(DEV)
    kw = {}
    if model == "controlled":
        with torch.no_grad():
            m(x, timesteps=t, context=ctx)  # shapes of the 13 control residuals: the skips + the middle output
        shapes = [(4, 64, 16, 16)] * 3 + [(4, 64, 8, 8)] + [(4, 128, 8, 8)] * 2 + [(4, 128, 4, 4)] + [(4, 256, 4, 4)] * 2 + [(4, 256, 2, 2)] * 4
        kw["control"] = [synth_input(f"cfgdup.ctrl{i}", sh, 7, 0.1).to(DEV) for i, sh in enumerate(shapes)]
    with torch.no_grad():
        full = m(x.clone(), timesteps=t, context=ctx, **kw)
        shared = m(ops.mark_cfg_dup(x.clone()), timesteps=t, context=ctx, **kw)
        shared_kw = m(x.clone(), timesteps=t, context=ctx, cfg_dup=True, **kw)
    assert torch.equal(shared, shared_kw)
    tol = TOL_BLOCK[dtype]
    close(shared, full.float().cpu(), tol, f"cfg-shared prefix vs full batch ({model})")
    assert not torch.equal(shared[:2], shared[2:])  # the halves differ (different conditioning)
    if model == "unet":
        gm = GraphedModule(m, scratch_bytes=64 << 20)
        with torch.no_grad():
            yg = gm(ops.mark_cfg_dup(x.clone()), timesteps=t, context=ctx)
            yg2 = gm(ops.mark_cfg_dup(x.clone()), timesteps=t, context=ctx)
            yu = gm(x.clone(), timesteps=t, context=ctx)  # unmarked: its own graph, the full-batch kernel sequence
        assert torch.equal(yg, shared) and torch.equal(yg2, shared) and torch.equal(yu, full) and len(gm._graphs) == 2
        x2 = torch.cat([xh, synth_input("cfgdup.x2", (2, 4, 16, 16), 8).to(DEV)])
        with torch.no_grad():
            y2 = m(x2, timesteps=t, context=ctx)
        assert not torch.allclose(y2[:2].float(), y2[2:].float())


def test_hip_graph_replay_equals_eager():
    """hipGraph replay of the UNet call (cremage_amd.graphs) is bitwise the eager result, for changing x / t and after a
    context change (re-capture), on the SD-shaped small UNet."""
    from cremage_amd.graphs import GraphedModule
    from cremage_amd.ldm_hip.unet import UNetModel
    meta, g = load_golden("unet_small_sd")
    cfg = meta["cfg"]
    m = prep(UNetModel(**cfg), meta, BF)
    gm = GraphedModule(m, scratch_bytes=64 << 20)
    ctx1 = synth_input("graph.ctx1", (4, 77, cfg["context_dim"]), 3).to(DEV)
    ctx2 = synth_input("graph.ctx2", (4, 77, cfg["context_dim"]), 4).to(DEV)
    with torch.no_grad():
        for step, ctx in enumerate([ctx1, ctx1, ctx1, ctx2, ctx2]):
            x = synth_input(f"graph.x{step}", (4, 4, 16, 16), 5).to(DEV)
            t = torch.full((4,), 900.0 - 100.5 * step, device=DEV)
            y_graph = gm(x, timesteps=t, context=ctx)
            y_eager = m(x, timesteps=t, context=ctx)
            assert torch.equal(y_graph, y_eager), step
    assert len(gm._graphs) == 2


@pytest.mark.gpu
def test_hip_graph_owns_its_kv_cache_across_context_switches():
    """ctx A -> ctx B -> ctx A: graph A is replayed after graph B's warm-up has overwritten the modules' single-slot K / V^T
    caches.  The graph record holds A's cache tensors, so the replay reads live memory (and not whatever the caching allocator
    put where K_A used to be); results stay bitwise equal to eager launches.  Also: rebuilding a packed weight image drops the
    graph (epoch) instead of replaying against a freed image."""
    import gc
    from cremage_amd.graphs import GraphedModule
    from cremage_amd.ldm_hip.unet import UNetModel
    meta, g = load_golden("unet_small_sd")
    cfg = meta["cfg"]
    m = prep(UNetModel(**cfg), meta, BF)
    gm = GraphedModule(m, scratch_bytes=64 << 20)
    ctxs = [synth_input(f"graph2.ctx{i}", (4, 77, cfg["context_dim"]), 7 + i).to(DEV) for i in range(2)]
    x = synth_input("graph2.x", (4, 4, 16, 16), 5).to(DEV)
    t = torch.full((4,), 700.5, device=DEV)
    with torch.no_grad():
        ref = [m(x, timesteps=t, context=c).clone() for c in ctxs]
        for rnd_ in range(3):
            for i in (0, 1, 0, 1):
                y = gm(x, timesteps=t, context=ctxs[i])
                # churn the allocator: anything freed by the other graph's warm-up gets reused and overwritten
                junk = [torch.full((1 << 18,), float(rnd_ + 1), device=DEV, dtype=torch.bfloat16) for _ in range(8)]
                del junk
                gc.collect()
                assert torch.equal(y, ref[i]), (rnd_, i)
        assert gm.captures == 2 and gm.replays == 10  # 12 calls, the two capturing ones are not replays
        # a rebuilt weight image (here: in-place weight edit -> _version bump -> repack on the next eager use) invalidates
        w = m.input_blocks[1][0].in_layers[2].weight
        w.mul_(1.0)  # same values, new version
        y = gm(x, timesteps=t, context=ctxs[0])
        assert torch.equal(y, ref[0]) and gm.captures == 3


def test_hip_graph_recaptures_after_weight_cache_clear():
    """ops.clear_weight_cache() frees the packed / stacked / fp32 images a captured graph baked in while no parameter changes:
    the cache generation is part of the graph's weight stamp, so the next call re-captures (and is right) instead of replaying
    against freed memory."""
    import gc
    from cremage_amd import ops
    from cremage_amd.graphs import GraphedModule
    from cremage_amd.ldm_hip.unet import UNetModel
    meta, g = load_golden("unet_small_sd")
    cfg = meta["cfg"]
    m = prep(UNetModel(**cfg), meta, BF)
    gm = GraphedModule(m, scratch_bytes=64 << 20)
    ctx = synth_input("graph3.ctx", (4, 77, cfg["context_dim"]), 9).to(DEV)
    x = synth_input("graph3.x", (4, 4, 16, 16), 5).to(DEV)
    t = torch.full((4,), 650.25, device=DEV)
    with torch.no_grad():
        ref = m(x, timesteps=t, context=ctx).clone()
        assert torch.equal(gm(x, timesteps=t, context=ctx), ref)
        assert torch.equal(gm(x, timesteps=t, context=ctx), ref)
        assert gm.captures == 1 and gm.replays == 1
        ops.clear_weight_cache()
        gc.collect()
        junk = [torch.full((1 << 20,), 3.0, device=DEV, dtype=torch.bfloat16) for _ in range(16)]  # reuse whatever was freed
        y = gm(x, timesteps=t, context=ctx)
        del junk
        assert gm.captures == 2, "a cache clear must invalidate the captured graph"
        assert torch.equal(y, ref)
        assert torch.equal(gm(x, timesteps=t, context=ctx), ref) and gm.replays == 2
