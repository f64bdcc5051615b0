"""Host-side glue either side of the denoising path: the auto-face-fix crop / pad / paste geometry, the
pixel-space Lanczos resizes and the PNG writer with the `generation_data` text chunk.

These steps run on the host on PIL images in the reference too (they are not part of the GPU hot path); they are restated here so
that a Cremage user finds the whole second-pass chain:

  face box -> crop plan            modules/face_detection/face_detector_engine.py:152-203   (buffer 20 px, clamp to the image, aspect-
                                   preserving resize to `target_edge_len`, centred padding on a white square)
  padded square -> img2img         (the UNet re-entry: cremage_amd.pipeline.img2img / img2img_sdxl)
  result -> un-pad, resize back    face_detector_engine.py:257-266
  paste                            face_detector_engine.py:268-288 uses cv.seamlessClone (NORMAL_CLONE); cv2 does not exist in this
                                   image, so Poisson blending is NOT restated (it could not be pinned against the reference) - the
                                   plain paste the reference keeps as a comment (:269) is what `paste_face` does, and says so
  PNG + generation_data            modules/sd/image_generator.py:1121-1212 (PngInfo.add_text("generation_data", json.dumps(...)))
  hires-fix pixel upscaler         image_generator.py:1020-1026 -> cremage/utils/ml_utils.py:28-71: cv2.resize(INTER_LANCZOS4) on uint8.
                                   cv2's 8x8 Lanczos-4 kernel is not PIL's Lanczos-3: `upscale_uint8` does the same uint8 round trip
                                   with PIL's filter and is labelled an approximation (unpinned).

Pure Python + PIL + numpy; nothing here touches the HIP library.
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import dataclass
from typing import Callable, Dict, Iterable, Optional, Sequence, Tuple

import numpy as np
import torch

FACE_BUFFER = 20  # face_detector_engine.py:158


@dataclass(frozen=True)
class FaceCropPlan:
    """Geometry of one face-fix pass.  (x, y, w, h): the crop rectangle in the base image after the buffer and the clamp;
    (new_w, new_h): the crop resized so that its longer edge is `edge`; (pad_x, pad_y): where it sits on the edge x edge square."""
    x: int
    y: int
    w: int
    h: int
    new_w: int
    new_h: int
    pad_x: int
    pad_y: int
    edge: int

    @property
    def crop_box(self) -> Tuple[int, int, int, int]:
        return (self.x, self.y, self.x + self.w, self.y + self.h)

    @property
    def inner_box(self) -> Tuple[int, int, int, int]:
        return (self.pad_x, self.pad_y, self.pad_x + self.new_w, self.pad_y + self.new_h)


def face_crop_plan(face: Sequence[float], image_size: Tuple[int, int], target_edge_len: int = 512, buffer: int = FACE_BUFFER) -> FaceCropPlan:
    """face = (x, y, w, h[, score]) as the detectors return it; image_size = PIL size (width, height).
    face_detector_engine.py:152-165 (int() of the box, grow by `buffer` on every side, clamp to the image) and :189-203 (landscape:
    width -> edge, height scaled and centred; otherwise height -> edge)."""
    x, y, w, h = int(face[0]), int(face[1]), int(face[2]), int(face[3])
    x = max(0, x - buffer)
    y = max(0, y - buffer)
    w = min(w + buffer * 2, image_size[0] - x)
    h = min(h + buffer * 2, image_size[1] - y)
    if w <= 0 or h <= 0:
        raise ValueError(f"face box {tuple(face[:4])} lies outside the {image_size[0]}x{image_size[1]} image")
    if w > h:  # landscape
        new_h = int(h * target_edge_len / w)
        new_w = target_edge_len
        pad_w, pad_h = 0, target_edge_len - new_h
    else:
        new_w = int(w * target_edge_len / h)
        new_h = target_edge_len
        pad_w, pad_h = target_edge_len - new_w, 0
    return FaceCropPlan(x, y, w, h, new_w, new_h, int(pad_w / 2), int(pad_h / 2), target_edge_len)


def crop_and_pad(image, plan: FaceCropPlan):
    """PIL image -> the white `edge` x `edge` RGBA square that goes to img2img (face_detector_engine.py:166-168,203-207)."""
    from PIL import Image
    crop = image.crop(plan.crop_box).convert("RGB")
    resized = crop.resize((plan.new_w, plan.new_h), resample=Image.LANCZOS)
    base = Image.new("RGBA", (plan.edge, plan.edge), "white")
    base.paste(resized, (plan.pad_x, plan.pad_y))
    return base


def unpad_and_resize(updated, plan: FaceCropPlan):
    """img2img output (edge x edge) -> the crop rectangle's size (face_detector_engine.py:257-266)."""
    from PIL import Image
    return updated.crop(plan.inner_box).resize((plan.w, plan.h), resample=Image.LANCZOS)


def paste_face(image, face_image, plan: FaceCropPlan, mode: str = "paste"):
    """Put the updated face back.  mode "paste": `pil_image.paste(updated, (x, y))`, the form the reference keeps as a comment
    (face_detector_engine.py:269).  mode "seamless" is the reference's live path (cv.seamlessClone NORMAL_CLONE, :271-286): cv2 is not
    installable here, so it is not restated."""
    if mode == "seamless":
        raise NotImplementedError("cv.seamlessClone (Poisson blending) needs OpenCV, which this build cannot import or pin against")
    if mode != "paste":
        raise ValueError(f"unknown paste mode {mode!r}")
    out = image.copy()
    out.paste(face_image.convert(image.mode), (plan.x, plan.y))
    return out


def pil_to_unit_tensor(image) -> torch.Tensor:
    """PIL RGB(A) -> float tensor [1, 3, H, W] in [-1, 1] (the img2img input convention, image_generator.py:697-706)."""
    a = np.asarray(image.convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1)[None] * 2.0 - 1.0


def unit_tensor_to_pil(x: torch.Tensor):
    """[3, H, W] in [0, 1] -> PIL RGB, the reference's `255. * x` -> uint8 truncation (image_generator.py:1151-1152)."""
    from PIL import Image
    a = (255.0 * x.detach().float().cpu().clamp(0, 1).permute(1, 2, 0).numpy()).astype(np.uint8)
    return Image.fromarray(a)


def face_fix(image, faces: Iterable[Sequence[float]], img2img_fn: Callable[[torch.Tensor], torch.Tensor], target_edge_len: int = 512,
             paste_mode: str = "paste"):
    """The auto-face-fix loop of one image (face_detector_engine.py:332-360 `fix_engine`: `process_face` for every detected face in turn, each
    pass working on the image the previous pass produced).  `img2img_fn`: [1, 3, edge, edge] in [-1, 1] -> [1, 3, edge, edge] in [0, 1] - the UNet
    re-entry (cremage_amd.pipeline.img2img / img2img_sdxl at the face-fix strength)."""
    for face in faces:
        plan = face_crop_plan(face, image.size, target_edge_len)
        square = crop_and_pad(image, plan)
        out = img2img_fn(pil_to_unit_tensor(square))
        if out.shape[-2:] != (plan.edge, plan.edge):
            raise ValueError(f"img2img returned {tuple(out.shape)} for a {plan.edge}x{plan.edge} input")
        image = paste_face(image, unpad_and_resize(unit_tensor_to_pil(out[0]), plan), plan, paste_mode)
    return image


def upscale_uint8(samples: torch.Tensor, width: int, height: int) -> torch.Tensor:
    """Pixel-space hires-fix upscaler: [b, c, h, w] in [0, 1] -> uint8 -> Lanczos resize -> back to [0, 1], the round trip of
    scale_pytorch_images (ml_utils.py:28-71).  APPROXIMATION: the reference's filter is cv2.INTER_LANCZOS4 (8x8 taps); PIL's LANCZOS
    is the 3-lobe kernel.  Shapes, dtype handling and the uint8 quantisation are the reference's, the tap weights are not."""
    from PIL import Image
    out = np.empty((samples.shape[0], height, width, samples.shape[1]), dtype=np.float32)
    u8 = (samples.detach().permute(0, 2, 3, 1) * 255.0).to(torch.uint8).cpu().numpy()
    for i in range(u8.shape[0]):
        out[i] = np.asarray(Image.fromarray(u8[i]).resize((width, height), resample=Image.LANCZOS), dtype=np.float32)
    return (torch.from_numpy(out).permute(0, 3, 1, 2) / 255.0).float().to(samples.device)


# ---------------------------------------------------------------------------------------------- PNG + generation_data
#: keys every image gets (image_generator.py:1123-1140), in the reference's order
GENERATION_KEYS = ("time", "positive_prompt", "negative_prompt", "ldm_model", "vae_model", "lora_models", "lora_weights", "sampler",
                   "sampling_iterations", "cfg", "image_height", "image_width", "clip_skip", "seed", "watermark", "safety_check")


def generation_parameters(*, positive_prompt: str, negative_prompt: str, ckpt: str, vae_ckpt: str, lora_models: str = "",
                          lora_weights: str = "", sampler: str = "Euler a", sampling_steps: int = 20, cfg: float = 7.5, height: int = 512,
                          width: int = 512, clip_skip: int = 1, seed: int = 0, image_index: int = 0, watermark: bool = False,
                          safety_check: bool = False, control_models: Optional[str] = None, face_input_img: Optional[str] = None,
                          face_strength: Optional[float] = None, hires_fix_upscaler: Optional[str] = None,
                          hires_fix_scale_factor: Optional[float] = None, auto_face_fix: bool = False,
                          auto_face_fix_strength: Optional[float] = None, auto_face_fix_prompt: Optional[str] = None,
                          auto_face_fix_face_detection_method: Optional[str] = None, now: Optional[float] = None) -> Dict:
    """The dict the reference serialises into the PNG (image_generator.py:1111-1147,1203-1207): model paths by basename, LoRA paths
    by basename joined with ',' (an empty LIST when there are none, :1111-1116), per-image seed = seed + index (:1135), and the
    conditional ControlNet / FaceID / hires-fix / face-fix keys."""
    if lora_models:
        loras = ",".join(os.path.basename(p) for p in lora_models.split(","))
    else:
        loras = []
    g = {"time": time.time() if now is None else now, "positive_prompt": positive_prompt, "negative_prompt": negative_prompt,
         "ldm_model": os.path.basename(ckpt), "vae_model": os.path.basename(vae_ckpt), "lora_models": loras, "lora_weights": lora_weights,
         "sampler": sampler, "sampling_iterations": sampling_steps, "cfg": cfg, "image_height": height, "image_width": width,
         "clip_skip": clip_skip, "seed": seed + image_index, "watermark": watermark, "safety_check": safety_check}
    if control_models:
        g["control_net"] = os.path.basename(control_models)
    if face_input_img:
        g["face_image"] = os.path.basename(face_input_img)
        g["face_strength"] = face_strength
    if hires_fix_upscaler and hires_fix_upscaler.lower() != "none":
        g["hires_fix_upscaler"] = hires_fix_upscaler
        g["hires_fix_scale_factor"] = hires_fix_scale_factor
        g["upscale_width"] = width * hires_fix_scale_factor
        g["upscale_height"] = height * hires_fix_scale_factor
    if auto_face_fix:
        g["auto_face_fix"] = True
        g["auto_face_fix_strength"] = auto_face_fix_strength
        g["auto_face_fix_prompt"] = auto_face_fix_prompt
        g["auto_face_fix_face_detection_method"] = auto_face_fix_face_detection_method
    return g


def save_png(image, params: Dict, directory: str, base_count: int = 0, now: Optional[float] = None) -> str:
    """`{base_count:05}_{time}.png` with ONE tEXt chunk, keyword "generation_data", value json.dumps(params)
    (image_generator.py:1209-1217).  `image`: PIL image or a [3, H, W] tensor in [0, 1].  Returns the path."""
    from PIL.PngImagePlugin import PngInfo
    if torch.is_tensor(image):
        image = unit_tensor_to_pil(image)
    meta = PngInfo()
    meta.add_text("generation_data", json.dumps(params))
    path = os.path.join(directory, f"{base_count:05}_{time.time() if now is None else now}.png")
    image.save(path, pnginfo=meta)
    return path


def read_generation_data(path: str) -> Dict:
    """What the reference's image list reads back: `pil_image.info["generation_data"]` (cremage/ui/image_listbox_handlers.py:225-228)."""
    from PIL import Image
    with Image.open(path) as im:
        return json.loads(im.info["generation_data"])
