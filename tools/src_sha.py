"""sha256 of every HIP source of the library, as JSON {file: sha}: stamped into the profile-derived files under profiles/ so that
bench.py can tell whether a committed PMC number still belongs to the kernel it is quoted for.
usage: python tools/src_sha.py                      -> prints the map
       python tools/src_sha.py --stamp a.json b.json -> adds / refreshes the key "hip_sources_sha" in the given JSON files"""
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha_map():
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "buffer-x_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "buffer-x_amd", "csrc", "*.h"))):
        out[os.path.basename(f)] = hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
    return out


if __name__ == "__main__":
    m = sha_map()
    if len(sys.argv) > 2 and sys.argv[1] == "--stamp":
        for p in sys.argv[2:]:
            d = json.load(open(p))
            d["hip_sources_sha"] = m
            json.dump(d, open(p, "w"), indent=1)
    else:
        print(json.dumps(m, indent=1))
