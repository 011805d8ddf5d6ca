"""Stamp of the kernel sources a profile was taken on: sha256 (first 16 hex digits) of every file under tf-mpc_amd/csrc.
The PMC summaries under profiles/ carry it (`csrc_sha16`), and bench.py only quotes a summary's HBM traffic while the
source of the kernel it belongs to still hashes to the recorded value -- a profile cannot silently outlive its kernel.
(The GPU box has no .git, so a commit hash cannot be taken where the profile is; the source hash can.)

python tools/source_stamp.py            -> JSON {file: sha16}"""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tf-mpc_amd", "csrc")


def stamp():
    out = {}
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, name), "rb") as fh:
                out[name] = hashlib.sha256(fh.read()).hexdigest()[:16]
    return out


def matches(recorded, files):
    """True iff every file of `files` hashes now to what `recorded` (a summary's csrc_sha16) holds."""
    if not recorded:
        return False
    now = stamp()
    return all(recorded.get(f) is not None and recorded.get(f) == now.get(f) for f in files)


if __name__ == "__main__":
    print(json.dumps(stamp(), indent=1))
