"""Facts derived from the one data asset the reference ships for this path, src/tf_smpl/smpl_faces.npy (the SMPL triangle list the
renderer uses): it pins the vertex count and index range the path's [B,6890,3] outputs must have.  Run here
(the reference is not available on the GPU box):  python tests/golden/make_topology_golden.py"""
import hashlib
import json
import os

import numpy as np

SRC = "/root/reference/src/tf_smpl/smpl_faces.npy"
f = np.load(SRC, allow_pickle=False)
edges = set()
for a, b, c in f.tolist():
    for u, v in ((a, b), (b, c), (c, a)):
        edges.add((min(u, v), max(u, v)))
out = {
    "source": "src/tf_smpl/smpl_faces.npy",
    "sha256": hashlib.sha256(open(SRC, "rb").read()).hexdigest(),
    "faces": int(f.shape[0]),
    "dtype": str(f.dtype),
    "min_index": int(f.min()),
    "max_index": int(f.max()),
    "distinct_vertices": int(len(np.unique(f))),
    "edges": len(edges),
    "euler_characteristic": int(len(np.unique(f)) - len(edges) + f.shape[0]),
}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "smpl_topology.json"), "w"), indent=1)
print(out)
