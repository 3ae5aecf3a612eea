"""Machine descriptions shipped with the library (constraints + interactions as data, `sp1_amd.machine` format).

`recursion.compress_machine()` builds the reference's recursion compress/shrink machine; `recursion_compress.json` is
its dump in the interchange format (regenerate with `python -m sp1_amd.machines.dump`), which is what a Rust-side
exporter would write for the RISC-V machine."""
import os

RECURSION_COMPRESS_JSON = os.path.join(os.path.dirname(os.path.abspath(__file__)), "recursion_compress.json")


def load_recursion_compress():
    from ..machine import load_machine
    with open(RECURSION_COMPRESS_JSON) as f:
        return load_machine(f)
