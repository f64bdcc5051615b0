"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/crg_hip.h
declares (no compute calls without a GPU), the ctypes binding covers them, the YAML `target:` plug point
instantiates the HIP classes, parameter names match the reference's checkpoint contract, and
the product path has no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch
import yaml

from tests.conftest import REPO, load_golden


def _declared():
    header = open(os.path.join(REPO, "include", "crg_hip.h")).read()
    return sorted(set(re.findall(r"\b(crg_[a-z0-9_]+)\s*\(", header)))


def test_library_exports_every_declared_symbol():
    from cremage_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for sym in _declared():
        assert hasattr(dll, sym), sym
    assert set(_declared()) == set(_lib.SIGNATURES), set(_declared()) ^ set(_lib.SIGNATURES)
    assert _lib.load().crg_version() == 103


def test_struct_layouts_match_header():
    """field order of the ctypes structures == field order of the C structs"""
    from cremage_amd import _lib
    header = open(os.path.join(REPO, "include", "crg_hip.h")).read()
    for cname, cls in [("crg_gemm_args", _lib.GemmArgs), ("crg_conv_args", _lib.ConvArgs)]:
        body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} " + cname + ";", header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            parts = [p.strip() for p in decl.split(",")]
            names.append(re.findall(r"(\w+)(?:\[\d+\])?$", parts[0])[0])   # (`int mx_log2[4]`: an array field counts once)
            names += [re.findall(r"(\w+)(?:\[\d+\])?$", p)[0] for p in parts[1:]]
        assert names == [f[0] for f in cls._fields_], (cname, names)


def test_no_cpu_fallback():
    from cremage_amd import _lib, ops
    with pytest.raises(_lib.CrgError):
        ops.linear(torch.zeros(4, 16), torch.zeros(8, 16))
    with pytest.raises(_lib.CrgError):
        ops.group_norm(torch.zeros(1, 32, 4, 4), torch.ones(32), torch.zeros(32), 32, 1e-5)
    with pytest.raises(_lib.CrgError):
        ops.attention(torch.zeros(1, 8, 64), torch.zeros(1, 8, 64), torch.zeros(1, 64, 8), 1, 8, 0.125)


def test_product_never_imports_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "cremage_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_yaml_plug_point_instantiates_hip_classes():
    """plug point 1: `target:` strings -> instantiate_from_config (ldm/util.py:81-96)"""
    from cremage_amd.ldm_hip.latent_diffusion import instantiate_from_config
    from cremage_amd.ldm_hip.unet import UNetModel
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    cfg = yaml.safe_load(open(os.path.join(REPO, "cremage_amd", "configs", "v1-inference-hip.yaml")))
    p = cfg["model"]["params"]
    assert p["unet_config"]["target"] == "cremage_amd.ldm_hip.unet.UNetModel"
    u = dict(p["unet_config"]["params"], model_channels=32, num_heads=4, context_dim=64)  # shrink: structure only
    unet = instantiate_from_config({"target": p["unet_config"]["target"], "params": u})
    assert isinstance(unet, UNetModel) and len(unet.input_blocks) == 12 and len(unet.output_blocks) == 12
    v = p["first_stage_config"]["params"]
    v = dict(v, ddconfig=dict(v["ddconfig"], ch=32))
    vae = instantiate_from_config({"target": p["first_stage_config"]["target"], "params": v})
    assert isinstance(vae, AutoencoderKL)


def test_parameter_names_match_reference_contract():
    """The 4 keys the reference's own test asserts (test/ldm/ldm_instantiation_test.py:21-25, under the
    `model.diffusion_model.` prefix) plus the key sample recorded from the reference's UNetModel."""
    from cremage_amd import pipeline as P
    from cremage_amd.ldm_hip.unet import UNetModel
    meta, _ = load_golden("unet_small_sd")
    sd = UNetModel(**meta["cfg"]).state_dict()
    for k in ["time_embed.0.weight", "input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight",
              "middle_block.1.transformer_blocks.0.attn2.to_k.weight", "out.2.weight"] + meta["key_sample"]:
        assert k in sd, k
    assert len(sd) == meta["n_keys"] == 686


def test_attention_mode_registry_and_lora_names():
    """plug point 2 (ATTENTION_MODES, attention.py:865-869) and the LoRA parameter naming scanned by
    `"_lora_" in name` at image_generator.py:408-453"""
    from cremage_amd.ldm_hip.transformer import BasicTransformerBlock, CrossAttention, SpatialTransformer
    assert set(BasicTransformerBlock.ATTENTION_MODES) >= {"softmax", "softmax-xformers", "softmax-original"}
    assert all(v is CrossAttention for v in BasicTransformerBlock.ATTENTION_MODES.values())
    st = SpatialTransformer(64, 2, 32, depth=1, context_dim=48, lora_ranks=[4, 8], lora_weights=[1.0, 0.5])
    names = [n for n, _ in st.named_parameters() if "_lora_" in n]
    for frag in ["proj_in_lora_downs.0.weight", "proj_out_lora_ups.1.weight", "attn1.q_lora_alphas.0", "attn2.v_lora_downs.1.weight",
                 "attn2.out_lora_ups.0.weight", "ff.net.0.proj_lora_downs.0.weight", "ff.net_2_lora_ups.1.weight"]:
        assert any(n.endswith(frag) for n in names), frag


def test_full_state_dict_contract_sha1():
    """sha1 over the sorted `name:shape` list of the reference's own SD1.5 UNet (plain, and with LoRA ranks [4, 16] +
    FaceID tokens) and AutoencoderKL == the same digest of the HIP drop-in classes: `load_state_dict`, the 792 LoRA keys
    (cremage/utils/sd15_weight_list_with_lora.py) and `to_k_ipa/to_v_ipa` land where the reference puts them."""
    import hashlib
    from cremage_amd.ldm_hip.unet import UNetModel
    from cremage_amd.ldm_hip.vae import AutoencoderKL
    meta, _ = load_golden("param_contract")

    def digest(m):
        items = sorted(f"{k}:{tuple(v.shape)}" for k, v in m.state_dict().items())
        return hashlib.sha1("\n".join(items).encode()).hexdigest(), len(items)

    with torch.device("meta"):
        u = UNetModel(**meta["unet_cfg"])
        ul = UNetModel(**dict(meta["unet_cfg"], lora_ranks=meta["lora_ranks"], lora_weights=meta["lora_weights"],
                              ipa_scale=meta["ipa_scale"], ipa_num_tokens=meta["ipa_num_tokens"]))
        ae = AutoencoderKL(meta["vae_dd"], None, 4)
    assert digest(u) == (meta["unet_sha1"], meta["unet_n"])
    assert digest(ul) == (meta["unet_lora_sha1"], meta["unet_lora_n"])
    assert digest(ae) == (meta["vae_sha1"], meta["vae_n"])


def test_reference_side_container_proof():
    """tests/golden/reference_container.json is written by oracle/check_reference_container.py in the build container: the
    REFERENCE's instantiate_from_config (ldm/util.py:81-96) built the reference's own LatentDiffusion / ControlLDM from this repo's
    *-hip.yaml files and the reference's `load_state_dict(sd, strict=False)` (image_generator.py:345) loaded reference-named keys
    into them.  Here (no reference needed): nothing was missing or unexpected, and the key-list digest recorded for the HIP classes
    inside that container is the digest of the classes as they are NOW (so the record cannot go stale silently)."""
    import hashlib
    import json
    rec = json.load(open(os.path.join(REPO, "tests", "golden", "reference_container.json")))

    def digest(sd):
        items = sorted(f"{k}:{tuple(v.shape)}" for k, v in sd.items())
        return hashlib.sha1("\n".join(items).encode()).hexdigest(), len(items)

    assert rec["sd15"]["container"] == "ldm.models.diffusion.ddpm.LatentDiffusion"
    assert rec["cldm"]["container"] == "cldm.cldm.ControlLDM"
    for tag in ("sd15", "cldm"):
        r = rec[tag]
        assert r["missing_keys"] == [] and r["unexpected_keys"] == [], (tag, r["missing_keys"][:5], r["unexpected_keys"][:5])
        assert r["sha1_ref"] == r["sha1_hip"] and r["n_keys_ref"] == r["n_keys_hip"]
        assert all(t.startswith("cremage_amd.") for t in r["hip_targets"]) and len(r["hip_targets"]) >= 2
    for part in ("network_config", "first_stage_config"):
        r = rec["sdxl"][part]
        assert r["missing_keys"] == [] and r["unexpected_keys"] == [] and r["sha1_ref"] == r["sha1_hip"], part
    # the HIP classes today, from the same YAML files, on the meta device (structure only)
    from cremage_amd.ldm_hip.latent_diffusion import instantiate_from_config
    cfgd = os.path.join(REPO, "cremage_amd", "configs")
    p15 = yaml.safe_load(open(os.path.join(cfgd, "v1-inference-hip.yaml")))["model"]["params"]
    pxl = yaml.safe_load(open(os.path.join(cfgd, "sd_xl_base-hip.yaml")))["model"]["params"]
    with torch.device("meta"):
        unet = instantiate_from_config(p15["unet_config"])
        vae = instantiate_from_config(p15["first_stage_config"])
        xl_unet = instantiate_from_config(pxl["network_config"])
        xl_vae = instantiate_from_config(pxl["first_stage_config"])
    assert digest(xl_unet.state_dict()) == (rec["sdxl"]["network_config"]["sha1_hip"], rec["sdxl"]["network_config"]["n_keys_hip"])
    assert digest(xl_vae.state_dict()) == (rec["sdxl"]["first_stage_config"]["sha1_hip"], rec["sdxl"]["first_stage_config"]["n_keys_hip"])
    # SD1.5 container keys = model.diffusion_model.* + first_stage_model.* (+ the schedule buffers of LatentDiffusion itself)
    n_model = len(unet.state_dict()) + len(vae.state_dict())
    assert 0 < rec["sd15"]["n_keys_hip"] - n_model < 40, (rec["sd15"]["n_keys_hip"], n_model)


def test_weight_cache_sees_versioned_writes_only():
    """pins the rule INTEGRATION.md states: the packed-weight cache is stamped with the Parameter's `_version`, so
    `p.copy_()` / `p.mul_()` / a fresh Parameter invalidate an entry, while a write through the `.data` alias does not (it bumps
    the alias's counter) and needs `ops.clear_weight_cache()`."""
    import torch
    from cremage_amd import ops
    cache = ops.TensorKeyedCache()
    p = torch.nn.Parameter(torch.zeros(4))
    cache.put((p,), "k", "packed-0")
    assert cache.get((p,), "k") == "packed-0"
    p.data.mul_(2.0)                       # invisible: the documented blind spot
    assert cache.get((p,), "k") == "packed-0"
    with torch.no_grad():
        p.copy_(torch.ones(4))             # visible: _version bump
    assert cache.get((p,), "k") is None
    cache.put((p,), "k", "packed-1")
    q = torch.nn.Parameter(torch.ones(4))  # LoRA-style replacement: another object
    assert cache.get((q,), "k") is None
    ops.clear_weight_cache()               # the public reset exists and is callable without a GPU
