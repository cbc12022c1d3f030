"""A canonical, hashable description of a built deploy net: graph (layers, types, bottoms, tops, parameter shapes), blob names and
shapes, outputs, and every numeric layer parameter that matters in a TEST-phase forward.  Used to pin the generated model zoo
(mscnn_amd/zoo.py) to the reference's shipped mscnn_deploy.prototxt files on boxes that do not have the reference checkout:
tests/golden/make_deploy_fingerprints.py computes it from the shipped files where they exist and commits the hashes,
tests/test_prototxt.py::test_generated_net_matches_the_committed_fingerprint_of_the_shipped_file recomputes it from the generator."""
import hashlib
import json


def _norm(v):
    if isinstance(v, dict):
        return {k: [_norm(x) for x in vs] for k, vs in sorted(v.items()) if k not in ("weight_filler", "bias_filler")}
    try:
        return float(v)
    except ValueError:
        return v.strip('"')


def describe(net):
    from oracle import pynet      # (test infrastructure: the prototxt text parser of the oracle's own net builder)
    layers = []
    for i, name in enumerate(net.layer_names):
        d = pynet.parse_param_text(net.layer_param_text(i))
        for k in ("name", "type", "bottom", "top", "param", "propagate_down", "phase"):
            d.pop(k, None)
        layers.append({"name": name, "type": net.layer_types[i], "bottoms": list(net.layer_bottoms(i)), "tops": list(net.layer_tops(i)),
                       "param_shapes": [list(s) for s in net.param_shapes(i)], "params": _norm(d)})
    return {"layers": layers, "blobs": {b: list(net.blob_shape(b)) for b in net.blob_names}, "blob_order": list(net.blob_names),
            "outputs": list(net.outputs)}


def fingerprint(net):
    return hashlib.sha256(json.dumps(describe(net), sort_keys=True).encode()).hexdigest()
