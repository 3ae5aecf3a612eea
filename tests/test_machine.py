"""The machine-description interchange format (sp1_amd/machine.py): dump -> JSON -> load reproduces the exact
constraint and interaction programs the prover consumes; malformed documents are rejected."""
import copy
import json

import numpy as np
import pytest

from shard_chips import make_shard_chips
from sp1_amd.machine import dump_machine, load_machine


def test_machine_roundtrip():
    chips, _ = make_shard_chips(5, 3, True, 2)
    doc = dump_machine([(a, i) for a, i, _, _ in chips])
    text = json.dumps(doc)
    loaded = load_machine(text)
    assert [i.name for _, i in loaded] == [c[1].name for c in chips]
    for (air, inter), (a0, i0, _, _) in zip(loaded, chips):
        assert np.array_equal(air.to_array(), a0.to_array()) and air.num_constraints == a0.num_constraints
        assert np.array_equal(inter.to_array(), i0.to_array())
        assert (air.main_width, air.prep_width) == (a0.main_width, a0.prep_width)


@pytest.mark.parametrize("mutate", [
    lambda d: d["chips"].reverse(),                                                  # not sorted by name
    lambda d: d["chips"][0]["constraints"].__setitem__(0, [0, 99, 0]),               # column out of range
    lambda d: d["chips"][0]["constraints"].__setitem__(3, [4, 3, 50]),               # forward reference
    lambda d: d["chips"][0]["sends"][0]["values"][0]["terms"].append(["main", 77, 1]),
    lambda d: d["chips"][1]["receives"][0]["multiplicity"].__setitem__("constant", 0x7F000001),
    lambda d: d.__setitem__("field", "BabyBear"),
])
def test_machine_rejects_malformed_documents(mutate):
    chips, _ = make_shard_chips(4, 1)
    doc = copy.deepcopy(dump_machine([(a, i) for a, i, _, _ in chips]))
    mutate(doc)
    with pytest.raises(ValueError):
        load_machine(json.dumps(doc))


def test_the_whole_trusted_mode_machine_and_the_wrap_machine_round_trip():
    """Every transcribed chip — the 62 supervisor-mode entries of the reference's cost table and the recursion wrap machine's nine —
    through the JSON interchange document and back: same programs word for word (HINT pseudo-instructions included), same
    interactions. This is the document a Rust exporter would write."""
    from sp1_amd.machines import recursion as RC
    from sp1_amd.machines import riscv as R
    from sp1_amd.machines import riscv_more as M
    names = sorted(set(R.CHIPS) | set(M.MORE_CHIPS))
    assert len(names) == 62
    for machine in ([R.chip(n) for n in names], RC.wrap_machine()):
        loaded = load_machine(json.dumps(dump_machine(machine)))
        assert [a.name for a, _ in loaded] == [a.name for a, _ in machine]
        for (air, inter), (a0, i0) in zip(loaded, machine):
            assert np.array_equal(air.to_array(), a0.to_array()) and air.num_constraints == a0.num_constraints, a0.name
            assert np.array_equal(inter.to_array(), i0.to_array()), a0.name
            assert (air.main_width, air.prep_width) == (a0.main_width, a0.prep_width)
