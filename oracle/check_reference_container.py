"""ORACLE tooling (build container only) - boundary proof FROM THE REFERENCE SIDE.

The drop-in claim of INTEGRATION.md is "edit the `target:` lines of the YAML and nothing else".  This script exercises
exactly that, with the reference's own code doing the work:

  1. the REFERENCE's `instantiate_from_config` (modules/ldm/util.py:81-96; sgm copy modules/sdxl/sgm/util.py) builds the
     reference's own containers - `ldm.models.diffusion.ddpm.LatentDiffusion` (v1-inference-hip.yaml), `cldm.cldm.ControlLDM`
     (cldm_v15-hip.yaml) - from this repo's *-hip.yaml files, i.e. with cremage_amd classes as unet / first stage / control
     model.  Only `cond_stage_config` (the CLIP text encoder, a network fetch) is replaced by torch.nn.Identity, as
     oracle/gen_golden.py does;
  2. the same container is built from the reference's ORIGINAL yaml (reference classes), and its `state_dict()` - the key
     names and shapes a real checkpoint has - is loaded into the HIP-class container with the reference's own call,
     `model.load_state_dict(sd, strict=False)` (modules/sd/image_generator.py:345).  strict=False silently DROPS mismatches, so
     the proof is that `missing_keys` and `unexpected_keys` come back EMPTY (and no size-mismatch error is raised);
  3. SDXL: `sgm.models.diffusion.DiffusionEngine` cannot be imported offline (its import chain needs kornia, open_clip and a real
     torchvision for transformers), so sd_xl_base-hip.yaml is proven at component level: the reference's sgm
     `instantiate_from_config` builds `network_config` / `first_stage_config` / `denoiser_config` from it, and the
     reference's own sgm UNetModel / Encoder+Decoder state-dict keys load into them with nothing missing or unexpected.

Result: tests/golden/reference_container.json (key counts, digests, the missing / unexpected lists), checked on CPU by
tests/test_abi_and_boundary.py without the reference.  No reference source is copied; it is imported and run.

Usage:  python oracle/check_reference_container.py [--only sd15 cldm sdxl]
"""
import argparse
import contextlib
import hashlib
import json
import os
import sys
import time

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["GPU_DEVICE"] = "cpu"

import torch  # noqa: E402
import yaml  # noqa: E402

import oracle.gen_golden as G  # noqa: E402  (installs the import-time stand-ins for absent third-party packages; reference on sys.path)

REF_CFG = "/root/reference/configs/ldm/configs/stable-diffusion"
HIP_CFG = os.path.join(REPO, "cremage_amd", "configs")
OUT = os.path.join(REPO, "tests", "golden", "reference_container.json")


def digest(sd):
    items = sorted(f"{k}:{tuple(v.shape)}" for k, v in sd.items())
    return hashlib.sha1("\n".join(items).encode()).hexdigest(), len(items)


def no_clip(model_cfg):
    """cond_stage_config names a CLIP checkpoint on the network; every other entry is used as written."""
    model_cfg = json.loads(json.dumps(model_cfg))
    model_cfg["params"]["cond_stage_config"] = {"target": "torch.nn.Identity"}
    return model_cfg


def targets(cfg, out=None):
    out = [] if out is None else out
    if isinstance(cfg, dict):
        if "target" in cfg:
            out.append(cfg["target"])
        for v in cfg.values():
            targets(v, out)
    return out


def prove_ldm(tag, hip_yaml, ref_yaml):
    from ldm.util import instantiate_from_config  # the reference's
    t0 = time.time()
    hip_cfg = no_clip(yaml.safe_load(open(os.path.join(HIP_CFG, hip_yaml)))["model"])
    ref_cfg = no_clip(yaml.safe_load(open(os.path.join(REF_CFG, ref_yaml)))["model"])
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        hip = instantiate_from_config(hip_cfg)
        ref = instantiate_from_config(ref_cfg)
    assert type(hip) is type(ref), (type(hip), type(ref))  # the reference's own container class in both cases
    hip_mods = sorted({type(m).__module__.split(".")[0] for m in hip.modules()})
    sd = ref.state_dict()  # reference-named keys, as a checkpoint has them
    r = hip.load_state_dict(sd, strict=False)  # image_generator.py:345
    d_ref, n_ref = digest(sd)
    d_hip, n_hip = digest(hip.state_dict())
    res = dict(container=f"{type(hip).__module__}.{type(hip).__name__}", hip_yaml=hip_yaml, ref_yaml=ref_yaml,
               hip_targets=[t for t in targets(hip_cfg) if t.startswith("cremage_amd")], top_level_packages=hip_mods,
               n_keys_ref=n_ref, n_keys_hip=n_hip, sha1_ref=d_ref, sha1_hip=d_hip, missing_keys=list(r.missing_keys),
               unexpected_keys=list(r.unexpected_keys), seconds=round(time.time() - t0, 1))
    print(f"[container] {tag}: {res['container']} with {res['hip_targets']}: {n_hip} keys, missing {len(r.missing_keys)}, "
          f"unexpected {len(r.unexpected_keys)}, digests equal: {d_ref == d_hip}")
    return res


def prove_sdxl():
    G._import_sgm()
    from sgm.util import instantiate_from_config  # the reference's sgm copy
    from sgm.modules.diffusionmodules import model as SV
    from sgm.modules.diffusionmodules import openaimodel as SU
    t0 = time.time()
    hip_cfg = yaml.safe_load(open(os.path.join(HIP_CFG, "sd_xl_base-hip.yaml")))["model"]["params"]
    ref_cfg = yaml.safe_load(open("/root/reference/modules/sdxl/configs/inference/sd_xl_base.yaml"))["model"]["params"]
    out = {}
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        unet = instantiate_from_config(hip_cfg["network_config"])
        ref_unet = SU.UNetModel(**ref_cfg["network_config"]["params"])
    r = unet.load_state_dict(ref_unet.state_dict(), strict=False)
    d_ref, n_ref = digest(ref_unet.state_dict())
    d_hip, n_hip = digest(unet.state_dict())
    out["network_config"] = dict(target=hip_cfg["network_config"]["target"], n_keys_ref=n_ref, n_keys_hip=n_hip, sha1_ref=d_ref, sha1_hip=d_hip,
                                 missing_keys=list(r.missing_keys), unexpected_keys=list(r.unexpected_keys))
    del ref_unet, unet
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        vae = instantiate_from_config(hip_cfg["first_stage_config"])
    dd = dict(ref_cfg["first_stage_config"]["params"]["ddconfig"], attn_type="vanilla")  # "vanilla-xformers" needs xformers; same parameters
    box = torch.nn.Module()
    box.encoder, box.decoder = SV.Encoder(**dd), SV.Decoder(**dd)
    box.quant_conv, box.post_quant_conv = torch.nn.Conv2d(2 * dd["z_channels"], 2 * 4, 1), torch.nn.Conv2d(4, dd["z_channels"], 1)
    r = vae.load_state_dict(box.state_dict(), strict=False)
    d_ref, n_ref = digest(box.state_dict())
    d_hip, n_hip = digest(vae.state_dict())
    out["first_stage_config"] = dict(target=hip_cfg["first_stage_config"]["target"], n_keys_ref=n_ref, n_keys_hip=n_hip, sha1_ref=d_ref,
                                     sha1_hip=d_hip, missing_keys=list(r.missing_keys), unexpected_keys=list(r.unexpected_keys),
                                     note="reference side = sgm Encoder + Decoder + the two 1x1 quant convs of AutoencoderKL (sgm/models/autoencoder.py "
                                          "needs pytorch_lightning + the sgm.modules package init, i.e. kornia / open_clip)")
    den = instantiate_from_config(hip_cfg["denoiser_config"])  # reference classes, unchanged in the hip yaml
    out["denoiser_config"] = dict(target=hip_cfg["denoiser_config"]["target"], cls=f"{type(den).__module__}.{type(den).__name__}",
                                  n_sigmas=int(den.sigmas.shape[0]))
    out["container"] = ("sgm.models.diffusion.DiffusionEngine not importable offline (kornia, open_clip, torchvision for transformers): "
                        "component-level proof")
    out["seconds"] = round(time.time() - t0, 1)
    print(f"[container] sdxl: network {out['network_config']['n_keys_hip']} keys missing {len(out['network_config']['missing_keys'])} "
          f"unexpected {len(out['network_config']['unexpected_keys'])}; first stage {out['first_stage_config']['n_keys_hip']} keys missing "
          f"{len(out['first_stage_config']['missing_keys'])} unexpected {len(out['first_stage_config']['unexpected_keys'])}")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=["sd15", "cldm", "sdxl"])
    a = ap.parse_args()
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    torch.set_grad_enabled(False)
    if "sd15" in a.only:
        res["sd15"] = prove_ldm("sd15", "v1-inference-hip.yaml", "v1-inference.yaml")
    if "cldm" in a.only:
        sys.path.insert(0, "/root/reference/modules")
        res["cldm"] = prove_ldm("cldm", "cldm_v15-hip.yaml", "cldm_v15.yaml")
    if "sdxl" in a.only:
        res["sdxl"] = prove_sdxl()
    json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    print(f"[container] wrote {OUT}")
