"""On-disk formats around the hot path -- SURVEY.md section 8f row 4.

  * write_est_log            the per-scene `est.log` files the reference's test loop appends for
                             3DMatch / 3DLoMatch (models/generic_reg_model.py:382-403)
  * read_trajectory / read_trajectory_info   Redwood `.log` / `.info` readers
                             (benchmark/benchmark_predator.py:82-153)
  * evaluate_registration / benchmark        the registration-recall protocol
                             (benchmark/benchmark_predator.py:222-375)
so that, given the dataset's gt.log / gt.info files, real-data accuracy of this package's poses
can be scored exactly like the reference's.  Host-side numpy: these are file formats and a
per-pair 6x6 quadratic form, not device work.  (The KITTI voxel pre-downsampling of the same
row is a device operator: ops.voxel_downsample.)
"""
import os
from collections import defaultdict

import numpy as np


def write_est_log(log_path: str, benchmark: str, batch: dict, pred: dict) -> None:
    """generic_reg_model.py:382-403: append every pair's predicted 4x4 pose to
    <log_path>/<benchmark>/<scene>/est.log as 'tgt_idx\\tsrc_idx\\t-1' + 4 tab-separated rows with
    12 decimals.  scene = second path component of src_path, indices from 'cloud_bin_<i>.pth'."""
    B = len(batch['src_xyz'])
    for b in range(B):
        scene = batch['src_path'][b].split(os.path.sep)[1]
        src_idx = int(os.path.basename(batch['src_path'][b]).split('_')[-1].replace('.pth', ''))
        tgt_idx = int(os.path.basename(batch['tgt_path'][b]).split('_')[-1].replace('.pth', ''))
        pose = pred['pose'][-1][b] if pred['pose'].ndim == 4 else pred['pose'][b]
        pose = np.asarray(pose.detach().cpu().numpy() if hasattr(pose, 'detach') else pose, dtype=np.float64)
        if pose.shape[0] == 3:
            pose = np.concatenate([pose, [[0., 0., 0., 1.]]], axis=0)
        scene_folder = os.path.join(log_path, benchmark, scene)
        os.makedirs(scene_folder, exist_ok=True)
        with open(os.path.join(scene_folder, 'est.log'), 'a') as fid:
            fid.write('{}\t{}\t{}\n'.format(tgt_idx, src_idx, -1))
            for i in range(4):
                fid.write('\t'.join(map('{0:.12f}'.format, pose[i])) + '\n')


def read_trajectory(filename: str, dim: int = 4):
    """benchmark_predator.py:82-119 -> (keys [n,3] str, traj [n,dim,dim])."""
    with open(filename) as f:
        lines = f.readlines()
    keys = [[t.strip() for t in ln.split('\t')[0:3]] for ln in lines[0::(dim + 1)]]
    rows = [ln.split('\t')[0:dim] for i, ln in enumerate(lines) if i % (dim + 1) != 0]
    traj = np.asarray(rows, dtype=np.float64).reshape(-1, dim, dim)
    return np.asarray(keys), traj


def read_trajectory_info(filename: str, dim: int = 6):
    """benchmark_predator.py:122-153 -> (n_fragments, info [n,6,6])."""
    with open(filename) as fid:
        contents = fid.readlines()
    n_pairs = len(contents) // 7
    assert len(contents) == 7 * n_pairs
    info, n_frame = [], 0
    for i in range(n_pairs):
        _, _, n_frame = [int(item) for item in contents[i * 7].strip().split()]
        info.append(np.stack([np.array(item.split(), dtype=np.float64) for item in contents[i * 7 + 1:i * 7 + 7]]))
    return n_frame, np.asarray(info, dtype=np.float64).reshape(-1, dim, dim)


def _mat2quat(R: np.ndarray) -> np.ndarray:
    """Unit quaternion (w, x, y, z), w >= 0, of a rotation matrix: the dominant eigenvector of
    the symmetric 4x4 K matrix -- the method of nibabel.quaternions.mat2quat, which the reference
    imports (benchmark_predator.py:14)."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = R.flat
    K = np.array([[Qxx - Qyy - Qzz, 0, 0, 0],
                  [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                  [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
                  [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


def compute_transformation_error(trans: np.ndarray, info: np.ndarray) -> float:
    """benchmark_predator.py:60-79: er = (t, q_xyz); er^T info er / info[0,0]."""
    er = np.concatenate([trans[:3, 3], _mat2quat(trans[:3, :3])[1:]])
    return float((er.reshape(1, 6) @ info @ er.reshape(6, 1)).item() / info[0, 0])


def evaluate_registration(num_fragment, result, result_pairs, gt_pairs, gt, gt_info, err2: float = 0.2):
    """benchmark_predator.py:222-282 (Redwood protocol; only non-consecutive pairs count)."""
    err2 = err2 ** 2
    gt_mask = np.zeros((num_fragment, num_fragment), dtype=np.int64)
    for idx in range(gt_pairs.shape[0]):
        i, j = int(gt_pairs[idx, 0]), int(gt_pairs[idx, 1])
        if j - i > 1:
            gt_mask[i, j] = idx
    n_gt = np.sum(gt_mask > 0)
    errors = np.full(result_pairs.shape[0], np.nan)
    good, n_res, flags = 0, 0, []
    for idx in range(result_pairs.shape[0]):
        i, j = int(result_pairs[idx, 0]), int(result_pairs[idx, 1])
        if gt_mask[i, j] > 0:
            n_res += 1
            gi = gt_mask[i, j]
            p = compute_transformation_error(np.linalg.inv(gt[gi]) @ result[idx], gt_info[gi])
            errors[idx] = p
            if p <= err2:
                good += 1
                flags.append(0)
            else:
                flags.append(1)
        else:
            flags.append(2)
    if n_res == 0:
        n_res += 1e6
    return good * 1.0 / n_res, good * 1.0 / n_gt, flags, errors


def _rotation_error_deg(R1, R2):
    e = (np.trace(np.swapaxes(R1, 1, 2) @ R2, axis1=1, axis2=2) - 1) / 2
    return np.degrees(np.arccos(np.clip(e, -1, 1)))


def benchmark(est_folder: str, gt_folder: str):
    """benchmark_predator.py:285-375 -> (report string, mean recall over scenes)."""
    scenes = sorted(os.listdir(gt_folder))
    short = ['Kitchen', 'Home 1', 'Home 2', 'Hotel 1', 'Hotel 2', 'Hotel 3', 'Study', 'MIT Lab']
    out = "Scene\t¦ prec.\t¦ rec.\t¦ re\t¦ te\t¦ samples\t¦\n"
    stats = defaultdict(list)
    precision, recall, n_valids = [], [], []
    for idx, scene in enumerate(scenes):
        gt_pairs, gt_traj = read_trajectory(os.path.join(gt_folder, scene, "gt.log"))
        n_valid = int(sum(abs(int(e[0]) - int(e[1])) > 1 for e in gt_pairs))
        n_valids.append(n_valid)
        n_frag, gt_info = read_trajectory_info(os.path.join(gt_folder, scene, "gt.info"))
        est_pairs, est_traj = read_trajectory(os.path.join(est_folder, scene, 'est.log'))
        p, r, flags, errors = evaluate_registration(n_frag, est_traj, est_pairs, gt_pairs, gt_traj, gt_info)
        ext = np.zeros((len(est_pairs), 4, 4))
        for ei, pair in enumerate(est_pairs):          # extract_corresponding_trajectors (:156-176)
            key = [pair[0], pair[1], gt_pairs[0][2]]
            ext[ei] = gt_traj[np.where((gt_pairs == key).all(axis=1))[0]]
        ok = np.array(flags) == 0
        re = _rotation_error_deg(ext[:, :3, :3], est_traj[:, :3, :3])[ok]
        te = np.linalg.norm(ext[:, :3, 3] - est_traj[:, :3, 3], axis=1)[ok]
        stats['re_median'].append(np.median(re) if len(re) else float('nan'))
        stats['te_median'].append(np.median(te) if len(te) else float('nan'))
        precision.append(p)
        recall.append(r)
        name = short[idx] if idx < len(short) else scene
        out += "{}\t¦ {:.3f}\t¦ {:.3f}\t¦ {:.3f}\t¦ {:.3f}\t¦ {:3d}¦\n".format(
            name, p, r, stats['re_median'][-1], stats['te_median'][-1], n_valid)
        np.save(os.path.join(est_folder, scene, 'flag.npy'), flags)
        np.save(os.path.join(est_folder, scene, 'errors.npy'), errors)
    wp = (np.array(n_valids) * np.array(precision)).sum() / max(np.sum(n_valids), 1)
    out += "Mean precision: {:.3f}: +- {:.3f}\n".format(np.mean(precision), np.std(precision))
    out += "Weighted precision: {:.3f}\n".format(wp)
    out += "Mean median RRE: {:.3f}: +- {:.3f}\n".format(np.mean(stats['re_median']), np.std(stats['re_median']))
    out += "Mean median RTE: {:.3F}: +- {:.3f}\n".format(np.mean(stats['te_median']), np.std(stats['te_median']))
    return out, float(np.mean(recall))
