"""Shared plumbing of the launchers: config, models, background, distributed bring-up."""
from __future__ import annotations

import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL peer mappings (must be set before the HIP runtime starts)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if PKG not in sys.path:
    sys.path.insert(0, PKG)

import nerf  # noqa: E402
from nerf import distributed as D  # noqa: E402


def init_distributed(backend: str = "nccl"):
    """One process per GPU (torchrun / torch.distributed.run).  Returns (rank, world, device).
    backend "nccl" is RCCL over xGMI (production); "gloo" lets several ranks share one GPU (tests: the collectives are
    staged through the host, the kernels and the launcher logic are the same).  NERFACE_DIST_FORCE=1: bring the process group
    up even at world size 1 (under torch.distributed.run --nproc-per-node 1) and run every collective of the N > 1 path --
    RCCL load, `device_id=` init, broadcast, the flat all-reduce on device memory -- on the one GPU a test box has."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("the MI355X `nerf` package needs a ROCm device")
    if backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if (world > 1 or D.force_collectives_requested()) and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group(backend="nccl", device_id=dev)      # nccl == RCCL on ROCm
        else:
            torch.distributed.init_process_group(backend=backend)
    return rank, world, dev


class FrameStager:
    """Host -> device staging of the one frame a training step needs (image, pose, expression) through a small pinned double
    buffer.  The dataset itself stays in pageable memory (a few thousand 512x512x3 float frames would otherwise be ~15 GB
    of page-locked memory per rank); the copy of step i+1 overlaps the kernels of step i."""

    def __init__(self, images, poses, expressions, device, slots: int = 2):
        self.images, self.poses, self.expressions, self.device = images, poses, expressions, device
        mk = lambda t: torch.empty(t.shape[1:], dtype=t.dtype).pin_memory()
        self.slots = [(mk(images), mk(poses), mk(expressions), torch.cuda.Event()) for _ in range(slots)]
        self.k = 0

    def fetch(self, idx: int):
        img, pose, expr, ev = self.slots[self.k]
        self.k = (self.k + 1) % len(self.slots)
        ev.synchronize()                                   # the transfer that last used this slot (2 steps ago) has landed
        img.copy_(self.images[idx])
        pose.copy_(self.poses[idx])
        expr.copy_(self.expressions[idx])
        out = (img.to(self.device, non_blocking=True), pose.to(self.device, non_blocking=True),
               expr.to(self.device, non_blocking=True))
        ev.record(torch.cuda.current_stream(self.device))
        return out


def load_config(path: str):
    with open(path, "r") as f:
        return nerf.CfgNode(yaml.load(f, Loader=yaml.FullLoader))


def build_models(cfg, device):
    """getattr(models, cfg.models.*.type)(...) exactly as the reference scripts construct them (TR:100-124)."""
    def make(m):
        return getattr(nerf.models, m.type)(
            num_encoding_fn_xyz=m.num_encoding_fn_xyz, num_encoding_fn_dir=m.num_encoding_fn_dir,
            include_input_xyz=m.include_input_xyz, include_input_dir=m.include_input_dir, use_viewdirs=m.use_viewdirs,
            num_layers=cfg.models.coarse.num_layers, hidden_size=cfg.models.coarse.hidden_size, include_expression=True).to(device)
    coarse = make(cfg.models.coarse)
    fine = make(cfg.models.fine) if hasattr(cfg.models, "fine") else None
    return coarse, fine


def build_encoders(cfg):
    c = cfg.models.coarse
    enc_xyz = nerf.get_embedding_function(num_encoding_functions=c.num_encoding_fn_xyz, include_input=c.include_input_xyz,
                                          log_sampling=c.log_sampling_xyz)
    enc_dir = nerf.get_embedding_function(num_encoding_functions=c.num_encoding_fn_dir, include_input=c.include_input_dir,
                                          log_sampling=c.log_sampling_dir) if c.use_viewdirs else None
    return enc_xyz, enc_dir


def load_background(basedir: str, H: int, W: int, device):
    """The fixed background the trainer conditions on (TR:159-168): basedir/bg/00050.png scaled to the image size."""
    from PIL import Image
    p = os.path.join(basedir, "bg", "00050.png")
    if not os.path.exists(p):
        return None
    im = Image.open(p)
    im.thumbnail((H, W))
    return torch.from_numpy(np.array(im).astype(np.float32) / 255.0)[..., :3].to(device)


class ScalarLog:
    """Rank-0 training log with the reference's TensorBoard tags (TR:203, 415-424, 518-541: train/loss, train/coarse_loss,
    train/fine_loss, train/psnr, validation/loss, validation/coarse_loss, validation/fine_loss, validation/psnr and the
    validation images): a torch.utils.tensorboard SummaryWriter when that imports (it needs the `tensorboard` package), else
    the same records as JSON lines in <logdir>/scalars.jsonl.  Values may be 0-d device tensors: they are kept on the device
    and read back together at flush(), which the trainer calls on the iterations that print anyway -- no per-iteration sync."""

    def __init__(self, logdir: str):
        self.pending = []
        self.writer = None
        self.path = os.path.join(logdir, "scalars.jsonl")
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer = SummaryWriter(logdir)
        except Exception:
            self.writer = None

    @property
    def kind(self) -> str:
        return "tensorboard" if self.writer is not None else "jsonl"

    def add_scalar(self, tag: str, value, step: int) -> None:
        self.pending.append((tag, value.detach() if torch.is_tensor(value) else float(value), int(step)))

    def add_image(self, tag: str, img_chw: torch.Tensor, step: int) -> None:
        if self.writer is not None:
            self.writer.add_image(tag, img_chw.detach().float().clamp(0, 1).cpu(), step)

    def flush(self) -> None:
        if not self.pending:
            return
        tens = [v for _, v, _ in self.pending if torch.is_tensor(v)]
        vals = iter(torch.stack([t.float().reshape(()) for t in tens]).cpu().tolist()) if tens else iter(())
        rows = [(tag, next(vals) if torch.is_tensor(v) else v, step) for tag, v, step in self.pending]
        self.pending = []
        if self.writer is not None:
            for tag, v, step in rows:
                self.writer.add_scalar(tag, v, step)
            self.writer.flush()
        else:
            import json
            with open(self.path, "a") as f:
                for tag, v, step in rows:
                    f.write(json.dumps({"tag": tag, "value": v, "step": step}) + "\n")

    def close(self) -> None:
        self.flush()
        if self.writer is not None:
            self.writer.close()
