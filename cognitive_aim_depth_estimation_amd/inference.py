"""Inference harness: the reference ``demo.py`` surface with a real batch loop.

Mirrors ``CognitiveAimInference`` of reference demo.py:26-600 (constructor,
``predict``, ``predict_batch``, tolerant checkpoint loading, EXIF handling,
preprocessing, heat-map rendering).  What changes is the loop of
demo.py:406-432, which runs one image at a time: here images are preprocessed,
stacked into batches, optionally sharded by image over the ranks of an
initialised ``torch.distributed`` group (one process per GPU) and the per-image
``[depth, confidence]`` pairs are all-gathered (RCCL on GPUs, gloo in tests).
"""
from __future__ import annotations

import math
import os
import sys
import warnings
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .model import create_model

try:                                                 # optional: zero-copy view of decoded pixels (see _decode)
    import pyarrow as _arrow
except Exception:                                    # pragma: no cover
    _arrow = None

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
DEFAULT_MODULES = ["ambient_stream", "iterative_focal_stream", "exif_prior_database"]
DEFAULT_EXIF = {"FocalLength": 50.0, "FNumber": 2.8, "ISOSpeedRatings": 100}


# --------------------------------------------------------------------------
# host-side pieces (no GPU needed)
# --------------------------------------------------------------------------
def load_config(config_path: str) -> dict:
    """YAML -> dict, with the two fix-ups of reference demo.py:39-53."""
    import yaml
    with open(config_path, "r", encoding="utf-8") as f:
        config = yaml.safe_load(f)
    config["debug_init"] = True
    if "cognitive_modules" not in config:
        config["cognitive_modules"] = list(DEFAULT_MODULES)
    return config


def camera_info_from_config(config: dict) -> Optional[dict]:
    """reference demo.py:56-65: 71 cameras unless (model.)exif_config.num_cameras says otherwise."""
    if "exif_prior_database" not in config.get("cognitive_modules", []):
        return None
    n = 71
    if "exif_config" in config:
        n = config["exif_config"].get("num_cameras", 71)
    elif "model" in config and "exif_config" in (config["model"] or {}):
        n = config["model"]["exif_config"].get("num_cameras", 71)
    return {"num_cameras": n}


def image_size_from_config(config: dict) -> Tuple[int, int]:
    """reference demo.py:154-160."""
    size = config["dataset"]["image_size"]
    if isinstance(size, int):
        return (size, size)
    if isinstance(size, list) and len(size) == 1:
        return (size[0], size[0])
    return tuple(size)


def preprocess_image(image, size: Tuple[int, int]) -> torch.Tensor:
    """Resize -> ToTensor -> Normalize of reference demo.py:162-166 without torchvision:
    PIL bilinear resize (what torchvision's Resize does for PIL input, antialias included),
    HWC uint8 -> CHW float / 255, ImageNet mean/std.  ``size`` is (height, width)."""
    from PIL import Image
    img = image.convert("RGB").resize((size[1], size[0]), Image.BILINEAR)
    arr = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).to(torch.float32) / 255.0
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    return (arr - mean) / std


class DevicePreprocessor:
    """``preprocess_image`` on the GPU: the decoded RGB bytes go to the device as they are (3 B per source
    pixel) and one C-ABI call resizes, scales and normalises them into a slot of the batch tensor,
    bit-identical to the host path (Pillow's integer bilinear resampler, reference demo.py:162-166).
    Per-axis weight tables are cached per (source length, output length)."""

    def __init__(self, size: Tuple[int, int], device):
        from . import tables
        self.size = (int(size[0]), int(size[1]))
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePreprocessor needs the GPU; use preprocess_image on the host")
        self._tables = tables
        self._axis = {}
        self._lut = tables.normalize_lut(IMAGENET_MEAN, IMAGENET_STD).contiguous().to(self.device)
        self._tmp = None

    def _axis_tables(self, n_in: int, n_out: int):
        key = (n_in, n_out)
        if key not in self._axis:
            bounds, coefs = self._tables.resample_coefficients(n_in, n_out)
            self._axis[key] = (torch.from_numpy(bounds).to(self.device), torch.from_numpy(coefs).to(self.device),
                               int(coefs.shape[1]))
        return self._axis[key]

    def __call__(self, image, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """image: PIL image (any mode) or uint8 array [h, w, 3].  out: optional fp32 [3, H, W] device view."""
        from ._lib import call, ptr, stream_ptr
        arr = np.array(image.convert("RGB") if hasattr(image, "convert") else image, dtype=np.uint8)
        if arr.ndim != 3 or arr.shape[2] != 3:
            raise ValueError(f"expected an RGB image, got array of shape {arr.shape}")
        h, w = int(arr.shape[0]), int(arr.shape[1])
        H, W = self.size
        src = torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
        xb, xc, xk = self._axis_tables(w, W)
        yb, yc, yk = self._axis_tables(h, H)
        need = h * W * 3
        if self._tmp is None or self._tmp.numel() < need:
            self._tmp = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty(3, H, W, dtype=torch.float32, device=self.device)
        if out.shape != (3, H, W) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != src.device:
            raise ValueError("out must be a contiguous fp32 [3, H, W] tensor on the preprocessing device")
        call("cogaim_preprocess_rgb8", ptr(src), h, w, w * 3, 3, ptr(xb), ptr(xc), xk, ptr(yb), ptr(yc), yk,
             ptr(self._tmp), ptr(self._lut), ptr(out), H, W, stream_ptr())
        return out


def _preprocess_batch(self, arrays, out: torch.Tensor, pool=None) -> torch.Tensor:
    """Several decoded images in one go: their bytes are packed into a pinned staging buffer (two, used in turn),
    cross to the device in ONE asynchronous copy on the current stream together with one 88-byte record per image
    (addresses, sizes, resampling tables), and the resize / normalise passes run as two launches over the whole batch
    (cogaim_preprocess_rgb8_batch).  arrays: uint8 [h, w, 3] (packed RGB) or [h, w, 4] (RGBX, fourth byte ignored) each, sizes
    may differ; out: fp32 [len(arrays), 3, H, W].
    pool: an executor whose threads do the packing copies side by side (the caller then waits once, not per image)."""
    from ._lib import call, ptr, stream_ptr
    H, W = self.size
    if out.shape[1:] != (3, H, W) or out.shape[0] < len(arrays) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous fp32 [n, 3, H, W] tensor")
    import time
    tm = getattr(self, "timing", None)
    t_0 = time.perf_counter()
    offs, total = [], 0
    for a in arrays:
        if a.ndim != 3 or a.shape[2] not in (3, 4) or a.dtype != np.uint8:
            raise ValueError(f"expected uint8 RGB / RGBX arrays, got {a.dtype} {a.shape}")
        offs.append(total)
        total += (a.size + 255) // 256 * 256
    rec_off = total                                # the per-image records ride behind the pixels in the same copy
    total += (len(arrays) * _ITEM.itemsize + 255) // 256 * 256
    st = getattr(self, "_stage", None)
    if st is None:
        st = self._stage = {"turn": 0, "pinned": [None, None], "event": [None, None], "dev": None}
    t = st["turn"] = 1 - st["turn"]
    if st["pinned"][t] is None or st["pinned"][t].numel() < total:
        st["pinned"][t] = torch.empty(max(total, 1 << 20), dtype=torch.uint8).pin_memory()
        st["event"][t] = None
    if st["event"][t] is not None:
        st["event"][t].synchronize()             # the copy that last read this staging buffer has finished
    if st["dev"] is None or st["dev"].numel() < total:
        st["dev"] = torch.empty(max(total, 1 << 20), dtype=torch.uint8, device=self.device)
    host = st["pinned"][t].numpy()
    tmp_offs, need = [], 0
    for a in arrays:
        tmp_offs.append(need)
        need += (int(a.shape[0]) * W * 3 + 255) // 256 * 256
    if self._tmp is None or self._tmp.numel() < need:
        self._tmp = torch.empty(need, dtype=torch.uint8, device=self.device)
    base, tmp_base = st["dev"].data_ptr(), self._tmp.data_ptr()
    rec = host[rec_off: rec_off + len(arrays) * _ITEM.itemsize].view(_ITEM)
    for j, (a, o) in enumerate(zip(arrays, offs)):
        h, w, px = int(a.shape[0]), int(a.shape[1]), int(a.shape[2])
        xb, xc, xk = self._axis_tables(w, W)
        yb, yc, yk = self._axis_tables(h, H)
        rec[j] = (base + o, w * px, ptr(xb), ptr(xc), ptr(yb), ptr(yc), tmp_base + tmp_offs[j], out[j].data_ptr(),
                  h, w, px, xk, yk, 0)
    t_1 = time.perf_counter()
    if pool is None:
        for a, o in zip(arrays, offs):
            np.copyto(host[o: o + a.size].reshape(a.shape), a)
    else:
        list(pool.map(lambda ao: np.copyto(host[ao[1]: ao[1] + ao[0].size].reshape(ao[0].shape), ao[0]), zip(arrays, offs)))
    t_2 = time.perf_counter()
    st["dev"][:total].copy_(st["pinned"][t][:total], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st["event"][t] = ev
    t_3 = time.perf_counter()
    if arrays:
        call("cogaim_preprocess_rgb8_batch", base + rec_off, len(arrays), max(int(a.shape[0]) for a in arrays),
             ptr(self._lut), H, W, stream_ptr())
    if tm is not None:
        t_4 = time.perf_counter()
        for k, v in (("stage_wait_s", t_1 - t_0), ("pack_s", t_2 - t_1), ("upload_issue_s", t_3 - t_2), ("launch_s", t_4 - t_3)):
            tm[k] = tm.get(k, 0.0) + v
    return out


# one image of cogaim_preprocess_rgb8_batch (88 bytes, the layout include/cogaim.h gives)
_ITEM = np.dtype([("src", "<u8"), ("row_bytes", "<i8"), ("xbounds", "<u8"), ("xcoefs", "<u8"), ("ybounds", "<u8"),
                  ("ycoefs", "<u8"), ("tmp", "<u8"), ("dst", "<u8"), ("h", "<i4"), ("w", "<i4"), ("px", "<i4"),
                  ("xk", "<i4"), ("yk", "<i4"), ("pad", "<i4")])
assert _ITEM.itemsize == 88
DevicePreprocessor.batch = _preprocess_batch


def extract_exif(image_path) -> Optional[Dict]:
    """reference demo.py:170-255: FocalLength / FNumber / ExposureTime / ISO with range checks, Make / Model.
    Takes a path or an already opened PIL image (the batch loop opens every file once)."""
    try:
        from PIL import Image
        from PIL.ExifTags import TAGS
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            image = Image.open(image_path) if isinstance(image_path, (str, os.PathLike)) else image_path
            try:
                raw = image._getexif()
            except Exception:
                try:
                    raw = dict(image.getexif().items()) or None
                except Exception:
                    raw = None
            if raw is None:
                return None

            def val(v):
                try:
                    if isinstance(v, tuple) and len(v) == 2:
                        return float(v[0]) / float(v[1])
                    return float(v) if isinstance(v, (int, float)) else float(str(v))
                except Exception:
                    return None
            out = {}
            for tag_id, v in raw.items():
                try:
                    tag = TAGS.get(tag_id, tag_id)
                    if tag == "FocalLength":
                        x = val(v)
                        if x and 10 <= x <= 500:
                            out[tag] = x
                    elif tag == "FNumber":
                        x = val(v)
                        if x and 1.0 <= x <= 32.0:
                            out[tag] = x
                    elif tag == "ExposureTime":
                        x = val(v)
                        if x and 0.0001 <= x <= 30.0:
                            out[tag] = x
                    elif tag == "ISOSpeedRatings":
                        x = val(v)
                        if x and 50 <= x <= 25600:
                            out[tag] = int(x)
                    elif tag in ("Make", "Model") and isinstance(v, str) and v.strip():
                        out[tag] = v.strip()
                except Exception:
                    continue
            return out or None
    except Exception as e:
        if "Corrupt EXIF" not in str(e):
            print(f"EXIF extraction failed: {e}")
        return None


def exif_rows(exifs: Sequence[Optional[Dict]], camera_to_id: Dict[str, int]) -> Dict[str, torch.Tensor]:
    """Batched form of reference demo.py:257-296 (defaults 50 mm, f/2.8, ISO 100, camera 0)."""
    fl, ap, iso, cam = [], [], [], []
    for e in exifs:
        e = e or {}
        fl.append(float(e.get("FocalLength", 50.0)))
        ap.append(float(e.get("FNumber", 2.8)))
        iso.append(float(e.get("ISOSpeedRatings", 100)))
        cam.append(int(camera_to_id.get(e.get("Model", "unknown"), 0)))
    return {"focal_length": torch.tensor(fl, dtype=torch.float32), "aperture": torch.tensor(ap, dtype=torch.float32),
            "iso": torch.tensor(iso, dtype=torch.float32), "camera_idx": torch.tensor(cam, dtype=torch.long)}


def host_thread_budget() -> Tuple[int, int]:
    """(decode threads, packer threads) of ONE rank's batch loop.  Every rank of a node runs its own loop, so the
    host's cores are divided by the ranks the launcher started on this node (LOCAL_WORLD_SIZE): eight ranks that
    each took 16 + 8 threads would put ~200 runnable threads on the cores of one host.  COGAIM_HOST_THREADS
    overrides the per-rank core share (boxes whose os.cpu_count() shows more cores than the job may use)."""
    try:
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    except ValueError:
        local_world = 1
    share = os.environ.get("COGAIM_HOST_THREADS")
    if share is not None:
        share = max(1, int(share))
    else:
        try:
            cpus = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            cpus = os.cpu_count() or 4
        share = max(1, cpus // local_world)
    return max(1, min(16, share)), max(1, min(8, share // 2))


def heatmap_postprocess(attn: torch.Tensor) -> torch.Tensor:
    """reference demo.py:533-552 for a whole batch on the device: cube, damp everything at or below the 70th
    percentile by 0.3, min-max normalise, lay out on the (zero padded) square grid.  attn [B,N] on the GPU ->
    fp32 [B,G,G] on the GPU."""
    from ._lib import call, stream_ptr
    a = attn.detach().to(torch.float32).contiguous()
    if a.dim() == 1:
        a = a[None]
    B, N = a.shape
    G = int(math.isqrt(N))
    if G * G != N:
        G += 1
    out = torch.empty(B, G, G, device=a.device, dtype=torch.float32)
    call("cogaim_heatmap_postprocess", a.data_ptr(), out.data_ptr(), B, N, G, stream_ptr())
    return out


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous image shard of rank ``rank``: sizes differ by at most one."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_shards(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather of the per-image rows of every rank ([n_local, C] -> [n_total, C], shard order).
    Shards are padded to a common length so that one collective suffices."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    per = (n_total + world - 1) // world
    buf = torch.zeros(per, local.shape[1], device=local.device, dtype=local.dtype)
    buf[: local.shape[0]] = local
    out = torch.empty(world * per, local.shape[1], device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(out, buf)
    rows = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        rows.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(rows, dim=0)


# --------------------------------------------------------------------------
# the harness
# --------------------------------------------------------------------------
class CognitiveAimInference:
    def __init__(self, config_path: str, checkpoint_path: str, device: str = "auto", batch_size: int = 32,
                 render: bool = True):
        if device == "auto":
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("this build runs the forward pass only on an MI355X (ROCm) device; "
                               "there is no CPU path (use the reference's own demo.py for CPU runs)")
        print(f"Using device: {self.device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.config = load_config(config_path)
        camera_info = camera_info_from_config(self.config)
        self.model = create_model(self.config, camera_info).to(self.device)
        print(f"Model status: use_ambient={self.model.use_ambient}, use_focal={self.model.use_focal}, "
              f"use_exif={self.model.use_exif}")
        self._load_checkpoint(checkpoint_path)
        self.model.eval()
        self.image_size = image_size_from_config(self.config)
        self.camera_to_id = {"unknown": 0}
        self.batch_size = int(batch_size)
        self.render = render
        self.preprocess = DevicePreprocessor(self.image_size, self.device)

    def _load_checkpoint(self, checkpoint_path: str):
        """Tolerant loader of reference demo.py:89-150: the state dict is ``ckpt['model_state_dict']``, else
        ``ckpt['main_model']``, else the file's top-level dict; entries whose name and shape match are loaded
        (strict=False); parameters the file does not supply are re-initialised as the reference does (:136-146:
        matrices xavier-uniform, vectors named *weight* normal(0, 0.01), *bias* zero); ANY failure keeps the
        initial weights.  Unlike the reference it only unpickles tensors (weights_only=True)."""
        print(f"Loading checkpoint: {checkpoint_path}")
        try:
            ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
            if "model_state_dict" in ckpt:
                state = ckpt["model_state_dict"]
            elif "main_model" in ckpt:
                state = ckpt["main_model"]
            else:
                state = ckpt
            own = self.model.state_dict()
            keep, skipped = {}, 0
            for k, v in state.items():
                if k in own and own[k].shape == v.shape:
                    keep[k] = v
                else:
                    skipped += 1
            self.model.load_state_dict(keep, strict=False)
            print(f"Successfully loaded {len(keep)}/{len(state)} parameters")
            if skipped:
                print(f"Skipped {skipped} incompatible parameters")
            missing = [k for k in own if k not in keep]
            if missing:
                print(f"Initialized {len(missing)} missing parameters with default values")
                with torch.no_grad():
                    for k in missing:                      # state_dict() tensors alias the parameters
                        if "weight" in k:
                            if own[k].dim() > 1:
                                torch.nn.init.xavier_uniform_(own[k])
                            else:
                                torch.nn.init.normal_(own[k], 0, 0.01)
                        elif "bias" in k:
                            torch.nn.init.zeros_(own[k])
                self.model.refresh_weights()
        except Exception as e:
            print(f"Error loading checkpoint: {e}")
            print("Continuing with randomly initialized weights...")

    # -- single image, same contract as reference demo.py:298-404 -----------------------------
    def predict(self, image_path: str, instruction: Optional[str] = None) -> Tuple[float, float, Dict]:
        res = self._run([image_path], [instruction])
        return res[0]

    # -- batch: same signature as reference demo.py:406-432, but batched and sharded ----------
    def predict_batch(self, image_paths: list, instructions: Optional[list] = None) -> list:
        if instructions is None:
            instructions = [None] * len(image_paths)
        elif len(instructions) != len(image_paths):
            raise ValueError("Number of instructions must match number of images")
        return self._run(list(image_paths), list(instructions), tolerate_errors=True)

    def _decode(self, path: str):
        """Host part of loading one file (worker threads): decoded pixels, EXIF, original size.
        Pixels come back as a uint8 [h, w, 4] VIEW of the decoder's own RGBX memory where Pillow can export it (Arrow
        C interface, images held in one block: up to 16 MB, i.e. 4 megapixels), else as a packed [h, w, 3] copy.  The
        copy is Python-level work under the interpreter lock (0.3 ms per 518 x 518 image against 0.015 ms for the
        view), which is what the 16 decode threads and the loop's two other threads compete for."""
        from PIL import Image
        opened = Image.open(path)
        exif_raw = extract_exif(opened)              # before convert(): the converted copy carries no EXIF
        image = opened if opened.mode == "RGB" else opened.convert("RGB")
        image.load()
        w, h = image.size
        if _arrow is not None and hasattr(image, "__arrow_c_array__"):
            try:
                flat = _arrow.array(image).flatten().to_numpy(zero_copy_only=True)       # keeps `image` alive
                if flat.size == h * w * 4:
                    return flat.reshape(h, w, 4), exif_raw, (w, h)
            except Exception:
                pass                                 # several blocks, or an export this Pillow / pyarrow pair lacks
        return np.asarray(image, dtype=np.uint8), exif_raw, (w, h)

    def _plan(self, idx: List[int], paths: List[str], instructions: List[Optional[str]]):
        """Requests of this rank -> forward calls.  A file that is asked for with several instructions is decoded
        once and its instructions go through ONE guided sweep (backbone and focal stream do not depend on the
        instruction, reference src/model.py:1257); files with the same instruction list share batches.
        Returns [(instruction tuple, [(path, [request index per instruction])])] in first-seen order."""
        per_path: Dict[str, Dict[Optional[str], List[int]]] = {}
        for i in idx:
            per_path.setdefault(paths[i], {}).setdefault(instructions[i], []).append(i)
        groups: Dict[tuple, list] = {}
        for path, by_ins in per_path.items():
            named = tuple(k for k in by_ins if k is not None)
            if None in by_ins:
                groups.setdefault((None,), []).append((path, [by_ins[None]]))
            if named:
                groups.setdefault(named, []).append((path, [by_ins[k] for k in named]))
        return list(groups.items())

    def _run(self, paths: List[str], instructions: List[Optional[str]], tolerate_errors: bool = False) -> list:
        """The batch loop.  Files are decoded by a thread pool; a producer thread uploads the bytes and runs the
        device preprocessing of chunk k+1 on a stream of its own, into one of three reused [B,3,H,W] buffers,
        while chunk k runs the forward; depth / confidence rows travel to pinned host memory asynchronously and
        are read one chunk later, so the loop has no per-chunk host-device synchronisation."""
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        import torch.distributed as dist
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        rank = dist.get_rank() if world > 1 else 0
        n = len(paths)
        lo, hi = shard_range(n, rank, world)
        results: List[Optional[tuple]] = [None] * n
        local = torch.full((hi - lo, 2), float("nan"))
        failure: List[BaseException] = []
        try:
            plan = self._plan(list(range(lo, hi)), paths, instructions)
            H, W = self.image_size
            chunks = []                      # (instruction tuple, [(path, request lists)])
            for ins, files in plan:
                per = self.batch_size if ins == (None,) or len(ins) == 1 else max(1, self.batch_size // len(ins))
                chunks += [(ins, files[s: s + per]) for s in range(0, len(files), per)]
            cap = max((len(c[1]) for c in chunks), default=1)
            kmax = max((len(c[0]) for c in chunks), default=1)
            slots = [{"x": torch.empty(cap, 3, H, W, device=self.device), "free": None,
                      "dc": torch.empty(kmax * cap * 2, dtype=torch.float32).pin_memory(),
                      "exif_f": torch.empty(3, cap, dtype=torch.float32).pin_memory(),
                      "exif_i": torch.empty(cap, dtype=torch.int64).pin_memory(),
                      "exif_f_dev": torch.empty(3, cap, dtype=torch.float32, device=self.device),
                      "exif_i_dev": torch.empty(cap, dtype=torch.int64, device=self.device)} for _ in range(4)]
            # Slot ownership is explicit: the producer takes a slot only from `free_slots`, the consumer puts it
            # back once the chunk's rows have been read (finish()), i.e. after the forward that read slot['x'] and
            # the copy into slot['dc'] have completed; slot['free'] is that forward's last event and the producer's
            # side stream waits on it before it overwrites the buffers.  Four slots: one finishing, one whose
            # forward is queued, two for the producer (one being filled, one waiting in `ready`).
            free_slots: "queue.Queue" = queue.Queue()
            for sl in slots:
                free_slots.put(sl)
            ready: "queue.Queue" = queue.Queue(maxsize=2)
            stop = threading.Event()           # set by the consumer when it gives up: the producer then ends
            side = torch.cuda.Stream(device=self.device)

            class _Stopped(Exception):
                pass

            def take_slot():
                while True:
                    if stop.is_set():
                        raise _Stopped()
                    try:
                        return free_slots.get(timeout=0.1)
                    except queue.Empty:
                        continue

            def hand_over(item):
                while True:
                    if stop.is_set():
                        raise _Stopped()
                    try:
                        ready.put(item, timeout=0.1)
                        return
                    except queue.Full:
                        continue

            timing = {"producer_wait_decode_s": 0.0, "producer_upload_preprocess_s": 0.0, "producer_wait_buffer_s": 0.0,
                      "consumer_wait_chunk_s": 0.0, "consumer_forward_issue_s": 0.0, "consumer_results_s": 0.0}
            self.last_timing = timing
            self.preprocess.timing = timing
            clock = __import__("time").perf_counter

            def produce():
                try:
                    n_decode, n_pack = host_thread_budget()
                    with ThreadPoolExecutor(max_workers=n_decode) as pool, \
                            ThreadPoolExecutor(max_workers=n_pack) as packers:
                        futs: Dict[int, list] = {}
                        for ci, (ins, files) in enumerate(chunks):
                            # decode at most three chunks ahead of the upload: decoded arrays (36 MB each for 12-MP
                            # sources) wait in host memory until their chunk is packed
                            for cj in range(ci, min(ci + 4, len(chunks))):
                                if cj not in futs:
                                    futs[cj] = [pool.submit(self._decode, f[0]) for f in chunks[cj][1]]
                            t0 = clock()
                            slot = take_slot()
                            if slot["free"] is not None:
                                side.wait_event(slot["free"])       # orders the overwrite behind the last reader
                                slot["free"].synchronize()          # the pinned rows are rewritten by the host
                                slot["free"] = None
                            timing["producer_wait_buffer_s"] += clock() - t0
                            good, arrays = [], []
                            t0 = clock()
                            for (path, reqs), fut in zip(files, futs.pop(ci)):
                                try:
                                    arr, exif_raw, size = fut.result()
                                    arrays.append(arr)
                                    good.append((path, reqs, exif_raw, size))
                                except Exception as e:
                                    if not tolerate_errors:
                                        raise ValueError(f"Cannot load image {path}: {e}")
                                    for r in reqs:
                                        for i in r:
                                            results[i] = (None, None, {"error": str(e)})
                            t1 = clock()
                            with torch.cuda.stream(side):
                                exif = None
                                if arrays:
                                    # uploaded here, not by the consumer, from pinned rows of the slot: a copy from
                                    # pageable memory makes the host wait for everything queued before it on its
                                    # stream (on the forward's stream: the previous chunk's whole forward)
                                    rows = exif_rows([g[2] for g in good], self.camera_to_id)
                                    nb = len(good)
                                    slot["exif_i"][:nb] = rows["camera_idx"]
                                    for r, k in enumerate(("focal_length", "aperture", "iso")):
                                        slot["exif_f"][r, :nb] = rows[k]
                                    slot["exif_i_dev"].copy_(slot["exif_i"], non_blocking=True)
                                    slot["exif_f_dev"].copy_(slot["exif_f"], non_blocking=True)
                                    exif = {"camera_idx": slot["exif_i_dev"][:nb]}
                                    for r, k in enumerate(("focal_length", "aperture", "iso")):
                                        exif[k] = slot["exif_f_dev"][r, :nb]
                                    self.preprocess.batch(arrays, slot["x"], pool=packers)
                                ev = torch.cuda.Event()
                                ev.record(side)
                            timing["producer_wait_decode_s"] += t1 - t0
                            timing["producer_upload_preprocess_s"] += clock() - t1
                            hand_over((ins, slot, good, ev, exif))
                except _Stopped:
                    pass
                except BaseException as e:            # surfaces in the consumer
                    failure.append(e)
                finally:
                    try:
                        hand_over(None)
                    except _Stopped:
                        pass

            torch.cuda.set_device(self.device)
            worker = threading.Thread(target=produce, daemon=True, name="cogaim-producer")
            worker.start()
            pending = None                     # previous chunk: its rows are read once the next one is queued

            def finish(item):
                ins, good, dc_host, ev, heat, K, imgs, slot = item
                try:
                    read_rows(ins, good, dc_host, ev, heat, K, imgs)
                finally:
                    free_slots.put(slot)       # its forward has completed and its rows have been read

            def read_rows(ins, good, dc_host, ev, heat, K, imgs):
                if hasattr(ev, "result"):            # a forward_async handle: collect, then bring the rows over
                    depth, conf = ev.result()
                    dc_host.copy_(torch.cat([depth, conf], dim=1)[None].float(), non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                ev.synchronize()
                for j, (path, reqs, exif_raw, size) in enumerate(good):
                    for k, name in enumerate(ins):
                        d, c = float(dc_host[k, j, 0]), float(dc_host[k, j, 1])
                        for i in reqs[k]:
                            local[i - lo, 0], local[i - lo, 1] = d, c
                            meta = {"image_path": paths[i], "original_size": size, "processed_size": (H, W),
                                    "exif_available": exif_raw is not None,
                                    "exif_source": "real" if exif_raw is not None else "default", "instruction": name,
                                    "cognitive_modules": self.config.get("cognitive_modules", []),
                                    "model_status": {"ambient": self.model.use_ambient, "focal": self.model.use_focal,
                                                     "exif": self.model.use_exif},
                                    "exif_data": exif_raw or dict(DEFAULT_EXIF, Model="Default", Make="Default")}
                            results[i] = (d, c, meta)
                            if self.render:
                                try:
                                    self._save_prediction_image(paths[i], imgs[j], d, c,
                                                                None if heat is None else heat[k][j], name)
                                except Exception as e:        # a failed drawing never costs the prediction
                                    print(f"Prediction image skipped for {paths[i]}: {e}")

            while True:
                t0 = clock()
                got = ready.get()
                timing["consumer_wait_chunk_s"] += clock() - t0
                if got is None:
                    break
                ins, slot, good, ev, exif = got
                t0 = clock()
                if good:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    b = len(good)
                    x = slot["x"][:b]
                    if hasattr(self.model, "_last_attention_weights"):
                        delattr(self.model, "_last_attention_weights")          # reference demo.py:334-335
                    if ins == (None,) and not self.render and not getattr(self.model, "faithful_curiosity", False):
                        # plain predictions, nothing to draw: two chunks in flight (the focal / tail part of one
                        # under the backbone of the next); the rows are collected when the next chunk is queued
                        handle = self.model.forward_async(x, exif)
                        done = handle.done_event
                        if done is None:
                            done = torch.cuda.Event()
                            done.record(cur)
                        slot["free"] = done
                        timing["consumer_forward_issue_s"] += clock() - t0
                        t0 = clock()
                        if pending is not None:
                            finish(pending)
                        timing["consumer_results_s"] += clock() - t0
                        pending = (ins, good, slot["dc"][: b * 2].view(1, b, 2), handle, None, 1, None, slot)
                        continue
                    if ins == (None,):
                        depth, conf = self.model(x, exif)
                        attn = self.model.get_attention_weights()
                        depth, conf = depth[None], conf[None]
                        attn = None if attn is None else attn[None]
                    elif len(ins) == 1:
                        depth, conf = self.model.forward_with_guidance(x, exif, ins[0])
                        attn = self.model.get_attention_weights()
                        depth, conf = depth[None], conf[None]
                        attn = None if attn is None else attn[None]
                    else:
                        depth, conf, attn = self.model.forward_with_guidance(x, exif, list(ins), return_attention=True)
                    K = len(ins)
                    dc_host = slot["dc"][: K * b * 2].view(K, b, 2)      # read by finish() before the slot comes round again
                    dc_host.copy_(torch.cat([depth, conf], dim=2).float(), non_blocking=True)
                    heat = imgs = None
                    if self.render:
                        imgs = x.cpu()
                        try:
                            if attn is not None:
                                hm = heatmap_postprocess(attn.reshape(K * b, -1)).cpu().numpy()
                                heat = hm.reshape(K, b, hm.shape[-2], hm.shape[-1])
                        except Exception as e:                # e.g. a grid beyond the heat-map kernel's limit
                            print(f"Heat map skipped: {e}")
                    done = torch.cuda.Event()
                    done.record(cur)
                    slot["free"] = done
                    timing["consumer_forward_issue_s"] += clock() - t0
                    t0 = clock()
                    if pending is not None:
                        finish(pending)
                    timing["consumer_results_s"] += clock() - t0
                    pending = (ins, good, dc_host, done, heat, K, imgs, slot)
                else:
                    free_slots.put(slot)       # every file of the chunk failed to load: nothing was queued on it
            if pending is not None:
                finish(pending)
            worker.join()
            if failure:
                raise failure[0]
            self.model.synchronize()
        except BaseException as e:
            # under torch.distributed every rank must still reach the collective below; this rank's
            # unfinished rows stay NaN there and the error is raised afterwards
            failure.append(e)
            # end the producer: it polls `stop` wherever it could block, so it neither stays parked on a full
            # queue nor keeps the device buffers and thread pools of this call alive
            if "stop" in locals():
                stop.set()
                try:
                    while True:
                        ready.get_nowait()
                except queue.Empty:
                    pass
                if "worker" in locals() and worker.is_alive():
                    worker.join(timeout=30)
        if world > 1:
            everyone = gather_shards(local.to(self.device), n).cpu()
            for i in range(n):
                if results[i] is None and not torch.isnan(everyone[i, 0]):
                    results[i] = (float(everyone[i, 0]), float(everyone[i, 1]),
                                  {"image_path": paths[i], "instruction": instructions[i], "rank": "remote"})
                elif results[i] is None:
                    results[i] = (None, None, {"error": "failed on a remote rank"})
        if failure:
            raise failure[0]
        return results

    def save_results(self, results: list, output_path: str):
        """Result list -> JSON file, reference demo.py:434-469 (which its own main() no longer calls, :686-687)."""
        import json

        def plain(o):
            if isinstance(o, dict):
                return {k: plain(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [plain(v) for v in o]
            if hasattr(o, "numerator") and hasattr(o, "denominator") and not isinstance(o, (int, bool)):
                return float(o.numerator) / float(o.denominator)
            try:
                json.dumps(o)
                return o
            except (TypeError, ValueError):
                return str(o)
        rows = [{"depth_value": None if d is None else float(d), "confidence_score": None if c is None else float(c),
                 "metadata": plain(m)} for d, c, m in results]
        with open(output_path, "w", encoding="utf-8") as f:
            json.dump(rows, f, indent=2, ensure_ascii=False)
        print(f"Results saved to: {output_path}")

    def _save_prediction_image(self, image_path, image_tensor, depth_value, confidence_score, grid, instruction):
        """Three-panel PNG of reference demo.py:471-600 (image | attention overlay | numbers); `grid` is the
        image's post-processed G x G heat map (heatmap_postprocess) or None."""
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            from scipy.ndimage import zoom
        except Exception as e:            # rendering is optional plumbing, never the measured path
            print(f"Prediction image skipped: {e}")
            return
        out_dir = Path("demo_results")
        out_dir.mkdir(exist_ok=True)
        stem = Path(image_path).stem
        out_path = out_dir / (f"{stem}_{instruction}_prediction.png" if instruction else f"{stem}_prediction.png")
        mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
        std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
        img = torch.clamp(image_tensor.cpu() * std + mean, 0, 1).permute(1, 2, 0).numpy()
        fig, axes = plt.subplots(1, 3, figsize=(18, 6))
        axes[0].imshow(img)
        axes[0].set_title("Original Image", fontsize=14, fontweight="bold")
        axes[0].axis("off")
        axes[1].imshow(img)
        if grid is not None:
            h, w = img.shape[:2]
            resized = zoom(grid, (h / grid.shape[0], w / grid.shape[1]), order=1)
            im = axes[1].imshow(resized, cmap="plasma", alpha=0.6, interpolation="bilinear", vmin=0, vmax=1)
            axes[1].set_title("Focus Map (Overlay)", fontsize=14, fontweight="bold")
            cbar = plt.colorbar(im, ax=axes[1], fraction=0.046, pad=0.04)
            cbar.set_label("Attention Weight", rotation=270, labelpad=15)
        else:
            axes[1].set_title("No Attention Data", fontsize=14)
        axes[1].axis("off")
        ax = axes[2]
        ax.text(0.5, 0.7, f"Predicted Depth: {depth_value:.4f}m", ha="center", va="center", transform=ax.transAxes,
                fontsize=16, fontweight="bold")
        ax.text(0.5, 0.5, f"Confidence: {confidence_score:.4f}", ha="center", va="center", transform=ax.transAxes,
                fontsize=14, color="blue")
        if instruction:
            ax.text(0.5, 0.3, f"Instruction: {instruction}", ha="center", va="center", transform=ax.transAxes,
                    fontsize=12)
        ax.set_title("Prediction Results")
        ax.axis("off")
        plt.tight_layout()
        plt.savefig(out_path, dpi=150, bbox_inches="tight")
        plt.close()
        print(f"Prediction image saved to: {out_path}")


def build_arg_parser():
    """Flags of reference demo.py:604-611 plus --batch_size / --no_render."""
    import argparse
    p = argparse.ArgumentParser(description="Cognitive-Aim Experiment B Inference (MI355X build)")
    p.add_argument("--config", type=str, default="configs/experiment_B.yaml", help="Configuration file path")
    p.add_argument("--checkpoint", type=str, default="checkpoints/cognitive_aim_model.pth", help="Model checkpoint path")
    p.add_argument("--image", type=str, help="Single image path")
    p.add_argument("--image_dir", type=str, help="Image directory path")
    p.add_argument("--instruction", type=str, help="Guidance instruction (center/left/right)")
    p.add_argument("--output", type=str, default="inference_results.json", help="Output file path")
    p.add_argument("--device", type=str, default="auto", help="Computing device (auto/cuda)")
    p.add_argument("--batch_size", type=int, default=32, help="images per forward call")
    p.add_argument("--no_render", action="store_true", help="skip the per-image PNG")
    return p


def list_images(image_dir: str) -> List[str]:
    """reference demo.py:654-662."""
    d = Path(image_dir)
    paths = []
    for ext in (".jpg", ".jpeg", ".png", ".bmp", ".tiff"):
        paths.extend(d.glob(f"*{ext}"))
        paths.extend(d.glob(f"*{ext.upper()}"))
    return [str(p) for p in sorted(set(paths))]


def main(argv=None):
    args = build_arg_parser().parse_args(argv)
    if not args.image and not args.image_dir:
        raise ValueError("Must specify either --image or --image_dir")
    if args.image and args.image_dir:
        raise ValueError("Cannot specify both --image and --image_dir")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:          # launched with torch.distributed.run: shard by image
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
        args.device = f"cuda:{os.environ.get('LOCAL_RANK', '0')}"
    print("Initializing Cognitive-Aim inference engine...")
    engine = CognitiveAimInference(args.config, args.checkpoint, args.device, args.batch_size, not args.no_render)
    if args.image:
        instruction = args.instruction if args.instruction else "center"        # reference demo.py:635
        depth, conf, meta = engine.predict(args.image, instruction)
        print("\n=== Inference Results ===")
        print(f"Image: {args.image}\nDepth value: {depth:.4f}\nConfidence: {conf:.4f}")
        print(f"Guidance instruction: {instruction}\nCognitive modules: {meta['cognitive_modules']}")
        print(f"Model status: {meta['model_status']}")
    else:
        paths = list_images(args.image_dir)
        if not paths:
            raise ValueError(f"No image files found in directory {args.image_dir}")
        print(f"Found {len(paths)} images")
        results = engine.predict_batch(paths)
        ok = [r for r in results if r[0] is not None]
        print(f"\n=== Batch Inference Results ===\nTotal images: {len(paths)}\nSuccessfully processed: {len(ok)}"
              f"\nFailed: {len(paths) - len(ok)}")
        if ok:
            depths = [r[0] for r in ok]
            print(f"Depth range: {min(depths):.4f} - {max(depths):.4f}\nAverage depth: {np.mean(depths):.4f}"
                  f"\nAverage confidence: {np.mean([r[1] for r in ok]):.4f}")
    print("\nInference completed!")
