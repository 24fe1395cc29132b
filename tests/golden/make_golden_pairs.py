"""tests/golden/make_golden_pairs.py -- pair list of the REAL ThreeDMatchDataset(split="test") (dataset/threedmatch.py, open3d and
nibabel stubbed: the constructor only parses gt.log files) over the two synthetic scenes of tests/golden/eval/ -> pairs.npz.
Run after make_golden_eval.py."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    for name in ("open3d", "nibabel", "nibabel.quaternions"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["nibabel"].quaternions = sys.modules["nibabel.quaternions"]
    sys.path.insert(0, REF)
    import dataset.threedmatch as TD
    # the reference hard-codes its 8 test scenes; the fixture tree holds two of them under <root>/test/3DMatch/gt_result
    root = os.path.join(HERE, "eval_root")
    os.makedirs(os.path.join(root, "test", "3DMatch"), exist_ok=True)
    link = os.path.join(root, "test", "3DMatch", "gt_result")
    if not os.path.exists(link):
        os.symlink(os.path.join(HERE, "eval", "gt_result"), link)
    cfg = types.SimpleNamespace(data=types.SimpleNamespace(root=root, benchmark="3DMatch"))
    src = open(os.path.join(REF, "dataset", "threedmatch.py")).read()
    have = sorted(os.listdir(link))
    ds = TD.ThreeDMatchDataset.__new__(TD.ThreeDMatchDataset)
    ds.config, ds.root, ds.split, ds.files, ds.length = cfg, root, "test", [], 0
    # run the reference method with its scene list restricted to the scenes present (the loop body is the reference's own)
    orig = TD.loadlog

    def guarded(p):
        return orig(p) if os.path.basename(p) in have else {}
    TD.loadlog = guarded
    ds.prepare_matching_pairs(split="test")
    TD.loadlog = orig
    np.savez_compressed(os.path.join(HERE, "eval", "pairs.npz"), files=np.array(ds.files), poses=np.array(ds.poses), root_after=ds.root)
    os.remove(link)
    os.removedirs(os.path.join(root, "test", "3DMatch"))
    print(len(ds.files), ds.files[0], ds.root)


if __name__ == "__main__":
    main()
