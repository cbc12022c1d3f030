#!/usr/bin/env python3
"""Writes tests/golden/deploy_fingerprints.json: the canonical fingerprint (tests/deploy_fingerprint.py) of every shipped
mscnn_deploy.prototxt the model zoo mirrors, computed from the REFERENCE'S OWN FILES.  Needs the reference checkout (run in the
build container): python tests/golden/make_deploy_fingerprints.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mscnn_amd import net as mnet, zoo          # noqa: E402
from deploy_fingerprint import fingerprint      # noqa: E402

REF = "/root/reference/examples"
out = {}
for model in sorted(zoo.MODELS):
    path = os.path.join(REF, zoo.MODELS[model][1])
    n = mnet.Net(path, device=-1)
    out[model] = {"file": zoo.MODELS[model][1], "layers": len(n.layer_names), "sha256": fingerprint(n)}
    print(model, out[model])
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "deploy_fingerprints.json"), "w"), indent=1, sort_keys=True)
