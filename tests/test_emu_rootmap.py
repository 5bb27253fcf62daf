"""The map objects of a replica (ptx_root_map: getRoot(), micromerge.ts:443-449, last-writer-wins per key :572-602) on the CPU
emulation of rootmap_core.h, against the reference's own answers (tests/golden/rootmap_ref.json, made by the type-erased reference:
tests/make_rootmap_golden.py) and — where node is installed — a live oracle run."""
import json
import os

import pytest

import helpers as H
from peritext_amd import abi, wire


def _golden():
    with open(os.path.join(H.GOLDEN, "rootmap_ref.json")) as f:
        g = json.load(f)
    assert g["impl"] == "ref" and g["docs"] == H.root_map_docs()
    return g["expected"]


@pytest.mark.parametrize("reverse", [0, 1, 2])
def test_root_maps_against_the_reference_fixture(reverse):
    H.check_root_maps(lambda b: H.emu_root_map(b, reverse=reverse), lambda b: H.emu_merge(b, reverse=reverse), _golden())


@pytest.mark.skipif(not H.have_node(), reason="node (oracle runtime) not installed")
def test_root_maps_against_a_live_oracle():
    H.check_root_maps(H.emu_root_map, H.emu_merge, H.oracle_apply(H.root_map_docs(), roots=True))


def test_text_only_logs_have_the_text_list_in_the_root():
    with open(os.path.join(H.GOLDEN, "ptxgen_mini.json")) as f:
        gen = json.load(f)
    batch = wire.encode_docs([d["logs"] for d in gen["docs"]])
    rm = H.emu_root_map(batch)
    assert (rm.logs["status"] == 0).all() and (rm.logs["n_entries"] == 1).all()
    assert all(wire.decode_root(batch, rm, l) == {"text": {"$list": True}} for l in range(batch.n_logs))


def test_more_map_ops_than_the_table_holds_is_a_capacity_status():
    docs = H.root_map_docs()
    batch = wire.encode_docs(docs[:1])
    rm = H.emu_root_map(batch, lds_bytes=16 + 3 * 40)
    assert (rm.logs["status"] == abi.ERR_CAPACITY).all() and (rm.logs["n_entries"] == 0).all()


def test_decode_changes_restores_the_map_ops():
    """wire.decode_changes is the inverse of encode_docs for the rows of the map objects too."""
    docs = H.root_map_docs()
    batch = wire.encode_docs(docs)
    log = 0
    for logs in docs:
        for changes in logs:
            assert wire.decode_changes(batch, log) == changes
            log += 1


@pytest.mark.parametrize("reverse", [0, 2])
def test_change_calls_on_map_objects(reverse):
    """Micromerge.change with InputOperations on map objects (makeMap / set / del / makeList with a path, micromerge.ts:400-425), mixed
    with text ops, two calls in a row, maps made earlier in the same call, the order-dependent CHILDREN table of the reference: the
    Changes the reference itself returned (rootmap_ref.json)."""
    with open(os.path.join(H.GOLDEN, "rootmap_ref.json")) as f:
        g = json.load(f)

    def change_fn(batch, ops):
        res = H.emu_merge(batch, admission=True, reverse=reverse)
        return H.emu_change(batch, res, ops, reverse=reverse)

    H.check_map_change_calls(change_fn, g["change"])
