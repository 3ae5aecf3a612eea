"""python -m sp1_amd.machines.dump: write recursion_compress.json from the transcription in recursion.py."""
import json

from . import RECURSION_COMPRESS_JSON
from .recursion import compress_machine
from ..machine import dump_machine

if __name__ == "__main__":
    doc = dump_machine(compress_machine())
    doc["source"] = "hand transcription of RecursionAir::compress_machine() (SP1 v6.4.0), see sp1_amd/machines/recursion.py"
    with open(RECURSION_COMPRESS_JSON, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", RECURSION_COMPRESS_JSON, sum(len(c["constraints"]) for c in doc["chips"]), "instructions")
