"""Dataset loader with the reference's `load_flame_data` interface (nerf/load_flame.py:40-211).

Host-side I/O, outside the timed path.  The reference reads PNGs with imageio and resizes with cv2
(neither is installed in the target image); this loader uses PIL + numpy and keeps the on-disk format:
`transforms_{train,val,test}.json` with `camera_angle_x`, `intrinsics` [fx, fy, cx_rel, cy_rel] and per
frame `file_path`, `transform_matrix` (4x4), `expression` (76) and optional `bbox` (4, relative).
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch


def _read_png(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def _area_resize(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_AREA) of a float32 image (H, W, C), as the reference calls it
    (LF:150-152, LF:171-175).  For an integer shrink factor -- half_res, the only resize a shipped config uses -- OpenCV
    averages each source block with one float accumulator in row-major order and multiplies by 1/area; that order is kept,
    so the pixels are the ones cv2 produces.  Other factors (the 25x25 `debug` thumbnails) fall back to PIL's box filter,
    which weights partially covered pixels like INTER_AREA but rounds differently."""
    h, w = img.shape[:2]
    if out_h > 0 and out_w > 0 and h % out_h == 0 and w % out_w == 0:
        fy, fx = h // out_h, w // out_w
        acc = np.zeros((out_h, out_w) + img.shape[2:], dtype=np.float32)
        for sy in range(fy):
            for sx in range(fx):
                acc = (acc + img[sy::fy, sx::fx]).astype(np.float32)
        return (acc * np.float32(1.0 / (fx * fy))).astype(np.float32)
    from PIL import Image
    chans = [np.asarray(Image.fromarray(img[..., c].astype(np.float32), mode="F").resize((out_w, out_h), Image.BOX))
             for c in range(img.shape[-1])]
    return np.stack(chans, axis=-1).astype(np.float32)


def translate_by_t_along_z(t):
    m = torch.eye(4)
    m[2][3] = t
    return m


def rotate_by_phi_along_x(phi):
    m = torch.eye(4)
    m[1, 1] = m[2, 2] = np.cos(phi)
    m[1, 2] = -np.sin(phi)
    m[2, 1] = -m[1, 2]
    return m


def rotate_by_theta_along_y(theta):
    m = torch.eye(4)
    m[0, 0] = m[2, 2] = np.cos(theta)
    m[0, 2] = -np.sin(theta)
    m[2, 0] = -m[0, 2]
    return m


def pose_spherical(theta, phi, radius):
    """Reference load_flame.py:32-37."""
    c2w = translate_by_t_along_z(radius)
    c2w = rotate_by_phi_along_x(phi / 180.0 * np.pi) @ c2w
    c2w = rotate_by_theta_along_y(theta / 180 * np.pi) @ c2w
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ c2w.numpy()
    return c2w


def load_flame_data(basedir, half_res=False, testskip=1, debug=False, expressions=True, load_frontal_faces=False,
                    load_bbox=True, test=False):
    """Returns (imgs, poses, render_poses, [H, W, intrinsics], i_split, expressions, frontal_imgs, bboxs) like the
    reference (debug mode returns the reference's shorter 6-tuple)."""
    splits = ["test"] if test else ["train", "val", "test"]
    metas = {}
    for s in splits:
        with open(os.path.join(basedir, f"transforms_{s}.json"), "r") as fp:
            metas[s] = json.load(fp)
    all_imgs, all_frontal, all_poses, all_expr, all_bbox, counts = [], [], [], [], [], [0]
    meta = None
    for s in splits:
        meta = metas[s]
        skip = 1 if (s == "train" or testskip == 0) else testskip
        imgs, frontal, poses, exprs, bboxs = [], [], [], [], []
        for frame in meta["frames"][::skip]:
            imgs.append(_read_png(os.path.join(basedir, frame["file_path"] + ".png")))
            if load_frontal_faces:
                frontal.append(_read_png(os.path.join(basedir, frame["file_path"] + "_frontal.png")))
            poses.append(np.array(frame["transform_matrix"]))
            exprs.append(np.array(frame["expression"]))
            if load_bbox:
                bboxs.append(np.array(frame["bbox"]) if "bbox" in frame else np.array([0.0, 1.0, 0.0, 1.0]))
        imgs = (np.array(imgs) / 255.0).astype(np.float32)
        if load_frontal_faces:
            frontal = (np.array(frontal) / 255.0).astype(np.float32)
        counts.append(counts[-1] + imgs.shape[0])
        all_imgs.append(imgs)
        all_frontal.append(frontal)
        all_poses.append(np.array(poses).astype(np.float32))
        all_expr.append(np.array(exprs).astype(np.float32))
        all_bbox.append(np.array(bboxs).astype(np.float32))
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(len(splits))]
    imgs = np.concatenate(all_imgs, 0)
    frontal_imgs = np.concatenate(all_frontal, 0) if load_frontal_faces else None
    poses = np.concatenate(all_poses, 0)
    exprs = np.concatenate(all_expr, 0)
    bboxs = np.concatenate(all_bbox, 0)
    H, W = imgs[0].shape[:2]
    focal = 0.5 * W / np.tan(0.5 * float(meta["camera_angle_x"]))
    intrinsics = np.array(meta["intrinsics"]) if meta.get("intrinsics") else np.array([focal, focal, 0.5, 0.5])
    render_poses = torch.stack([torch.from_numpy(pose_spherical(a, -30.0, 4.0)) for a in np.linspace(-180, 180, 41)[:-1]], 0)

    def _stack(arr, size=None):
        if size is None:
            return torch.stack([torch.from_numpy(a) for a in arr], 0)
        return torch.stack([torch.from_numpy(_area_resize(a, size[0], size[1])) for a in arr], 0)

    if debug:
        H, W = H // 32, W // 32
        intrinsics[:2] = intrinsics[:2] / 32.0
        imgs_t = _stack(imgs, (25, 25))
        fr_t = _stack(frontal_imgs, (25, 25)) if frontal_imgs is not None else None
        return imgs_t, torch.from_numpy(poses), render_poses, [H, W, intrinsics], i_split, fr_t
    if half_res:
        H, W = H // 2, W // 2
        intrinsics[:2] = intrinsics[:2] * 0.5
        imgs_t = _stack(imgs, (H, W))
        fr_t = _stack(frontal_imgs, (H, W)) if load_frontal_faces else frontal_imgs
    else:
        imgs_t = _stack(imgs)
        fr_t = _stack(frontal_imgs) if load_frontal_faces else frontal_imgs
    bboxs[:, 0:2] *= H
    bboxs[:, 2:4] *= W
    bboxs_t = torch.from_numpy(np.floor(bboxs)).int()
    return (imgs_t, torch.from_numpy(poses), render_poses, [H, W, intrinsics], i_split, torch.from_numpy(exprs), fr_t, bboxs_t)
