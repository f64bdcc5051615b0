"""Host-side glue of SURVEY.md 8f row 4 (cremage_amd/postprocess.py): crop / pad / paste geometry of the auto-face-fix
(modules/face_detection/face_detector_engine.py:152-288) and the PNG `generation_data` writer (modules/sd/image_generator.py:1111-1217).
The reference code for these steps sits inside functions that need OpenCV / a downloaded ViT and cannot be imported here, so the
expectations below are worked out by hand from the cited lines (integer arithmetic and a JSON text chunk), not captured."""
import json

import numpy as np
import pytest
import torch

from cremage_amd import postprocess as PP


@pytest.mark.parametrize("face,size,edge,want", [
    # (x, y, w, h) box, image (W, H), target edge -> (x, y, w, h, new_w, new_h, pad_x, pad_y)
    # portrait box well inside: grow by 20 on each side; h >= w -> height to the edge, width scaled by int(), padding centred by int(/2)
    ((100, 120, 60, 90), (512, 512), 512, (80, 100, 100, 130, int(100 * 512 / 130), 512, int((512 - int(100 * 512 / 130)) / 2), 0)),
    # landscape box
    ((200.7, 50.2, 180, 100), (1024, 1024), 1024, (180, 30, 220, 140, 1024, int(140 * 1024 / 220), 0, int((1024 - int(140 * 1024 / 220)) / 2))),
    # clamped at the top-left corner: x, y stop at 0, w / h still grow by 40 (face_detector_engine.py:159-162)
    ((5, 8, 50, 50), (256, 256), 512, (0, 0, 90, 90, 512, 512, 0, 0)),
    # clamped at the bottom-right: w, h are cut to what is left of the image
    ((200, 210, 60, 60), (256, 256), 512, (180, 190, 76, 66, 512, int(66 * 512 / 76), 0, int((512 - int(66 * 512 / 76)) / 2))),
])
def test_face_crop_plan(face, size, edge, want):
    p = PP.face_crop_plan(face, size, edge)
    assert (p.x, p.y, p.w, p.h, p.new_w, p.new_h, p.pad_x, p.pad_y) == want
    assert p.crop_box == (p.x, p.y, p.x + p.w, p.y + p.h)
    ib = p.inner_box
    assert 0 <= ib[0] and ib[2] <= edge and 0 <= ib[1] and ib[3] <= edge


def test_face_crop_plan_rejects_boxes_outside():
    with pytest.raises(ValueError):
        PP.face_crop_plan((600, 10, 40, 40), (512, 512), 512)


def test_face_fix_round_trip_geometry_and_paste():
    """crop -> white-padded square -> (identity "img2img") -> un-pad -> resize back -> paste: the image outside the crop rectangle is
    untouched, the square is white outside the resized crop, the pasted region keeps the crop's content up to two Lanczos resamplings."""
    from PIL import Image
    rng = np.random.default_rng(3)
    base = np.kron(rng.integers(0, 255, (16, 20, 3), dtype=np.uint8), np.ones((16, 16, 1), dtype=np.uint8))  # 256 x 320 blocky image
    img = Image.fromarray(base)
    face = (90, 60, 70, 100, 0.99)
    plan = PP.face_crop_plan(face, img.size, 512)
    sq = PP.crop_and_pad(img, plan)
    assert sq.size == (512, 512) and sq.mode == "RGBA"
    a = np.asarray(sq.convert("RGB"))
    assert (a[:, :plan.pad_x] == 255).all() and (a[:, plan.pad_x + plan.new_w:] == 255).all()
    seen = {}

    def identity(x):
        seen["shape"] = tuple(x.shape)
        assert x.min() >= -1 and x.max() <= 1
        return (x + 1) * 0.5
    out = PP.face_fix(img, [face], identity, 512)
    assert seen["shape"] == (1, 3, 512, 512) and out.size == img.size
    o = np.asarray(out.convert("RGB")).astype(np.int32)
    mask = np.ones(o.shape[:2], bool)
    mask[plan.y:plan.y + plan.h, plan.x:plan.x + plan.w] = False
    assert (o[mask] == base.astype(np.int32)[mask]).all()
    inner = np.abs(o[~mask] - base.astype(np.int32)[~mask])
    assert inner.mean() < 12  # blocky content through an up- and a down-sampling by Lanczos
    with pytest.raises(NotImplementedError):
        PP.paste_face(img, img.crop(plan.crop_box), plan, mode="seamless")


def test_generation_data_png_round_trip(tmp_path):
    params = PP.generation_parameters(positive_prompt="a cat", negative_prompt="blurry", ckpt="/m/sd15/model.safetensors", vae_ckpt="/m/vae/v.pt",
                                      lora_models="/l/a.safetensors,/l/b.safetensors", lora_weights="1.0,0.5", sampler="Euler a",
                                      sampling_steps=20, cfg=7.5, height=512, width=768, clip_skip=2, seed=42, image_index=3,
                                      hires_fix_upscaler="Latent", hires_fix_scale_factor=2.0, auto_face_fix=True, auto_face_fix_strength=0.3,
                                      auto_face_fix_prompt="a cat", auto_face_fix_face_detection_method="InsightFace", now=1700000000.5)
    assert tuple(params)[:len(PP.GENERATION_KEYS)] == PP.GENERATION_KEYS  # the reference's key order (image_generator.py:1123-1140)
    assert params["ldm_model"] == "model.safetensors" and params["vae_model"] == "v.pt" and params["lora_models"] == "a.safetensors,b.safetensors"
    assert params["seed"] == 45 and params["upscale_width"] == 1536 and params["upscale_height"] == 1024
    assert PP.generation_parameters(positive_prompt="", negative_prompt="", ckpt="c", vae_ckpt="v")["lora_models"] == []  # :1116
    img = torch.rand(3, 32, 48)
    path = PP.save_png(img, params, str(tmp_path), base_count=7, now=1700000001.25)
    assert path.endswith("00007_1700000001.25.png")
    assert PP.read_generation_data(path) == json.loads(json.dumps(params))
    raw = open(path, "rb").read()
    i = raw.index(b"tEXt")
    assert raw[i + 4:i + 4 + len(b"generation_data\x00")] == b"generation_data\x00"  # ONE uncompressed text chunk, keyword as the reference's
    assert raw.count(b"tEXt") == 1
    from PIL import Image
    with Image.open(path) as im:
        assert im.size == (48, 32)
        assert (np.asarray(im) == (255.0 * img.permute(1, 2, 0).numpy()).astype(np.uint8)).all()  # `255. * x` then uint8 truncation (:1151-1152)


def test_upscale_uint8_shapes_and_range():
    x = torch.rand(2, 3, 16, 24)
    y = PP.upscale_uint8(x, width=48, height=32)
    assert y.shape == (2, 3, 32, 48) and y.dtype == torch.float32 and 0.0 <= y.min() and y.max() <= 1.0
    flat = torch.full((1, 3, 8, 8), 100 / 255.0)
    assert torch.allclose(PP.upscale_uint8(flat, 16, 16), torch.full((1, 3, 16, 16), 100 / 255.0))
