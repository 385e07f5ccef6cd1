"""tests/golden/shipped_recipes.json: the recipes the reference ships for the paths in scope, as DATA.

TEST INFRASTRUCTURE ONLY (build container: reads /root/reference).  Usage:  python -m oracle.make_golden_options

For codes/options/sr/train_sr.yml, sr/train_sr.json, sr/test_sr.yml, i2i/train_pix2pix.yml and i2i/train_cyclegan.yml the fixture
holds the option tree the file denotes -- every key and value in file order, read the way the reference reads them
(options/options.py:539-560: ordered mappings, `5e-3`-style scalars as floats, `//` comments stripped from JSON) -- with the
filesystem locations (every string that starts with ../: dataset folders, path.root, pretrained models) replaced by the
placeholder @ROOT@/, which a test substitutes with a temporary directory it populates.  Comments and layout are not kept: the
fixture is the recipe's content, not a copy of the file.  tests/fixtures `write_recipe` turns an entry back into a .yml / .json
file for `options.parse`; tests/test_cpu_host.py::test_shipped_recipe_fixture_is_the_reference_files checks, where the reference
checkout is present, that the real files parse to exactly these trees.
"""
import json
import os
import re
from collections import OrderedDict

from .ref_harness import REF_CODES

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
OUT = os.path.join(GOLDEN, "shipped_recipes.json")
RECIPES = ("sr/train_sr.yml", "sr/train_sr.json", "sr/test_sr.yml", "i2i/train_pix2pix.yml", "i2i/train_cyclegan.yml")

_FLOAT = re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?|[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
    |\.[0-9_]+(?:[eE][-+]?[0-9]+)?|[-+]?\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$""", re.X)


def read_recipe(path):
    """The tree a recipe file denotes (ordered; YAML 1.2-style floats like the reference's loader resolves them)."""
    if path.endswith(".json"):
        with open(path) as f:
            return json.loads("\n".join(line.split("//")[0] for line in f), object_pairs_hook=OrderedDict)
    import yaml

    class Loader(yaml.SafeLoader):
        pass

    Loader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG, lambda l, n: OrderedDict(l.construct_pairs(n)))
    Loader.add_implicit_resolver("tag:yaml.org,2002:float", _FLOAT, list("-+0123456789."))
    with open(path) as f:
        return yaml.load(f, Loader=Loader)


def reroot(node):
    if isinstance(node, dict):
        return OrderedDict((k, reroot(v)) for k, v in node.items())
    if isinstance(node, list):
        return [reroot(v) for v in node]
    if isinstance(node, str) and node.startswith("../"):
        return "@ROOT@/" + node[3:]
    return node


def reference_trees():
    return OrderedDict((rel, reroot(read_recipe(os.path.join(REF_CODES, "options", rel)))) for rel in RECIPES)


def main():
    trees = reference_trees()
    with open(OUT, "w") as f:
        json.dump(trees, f, indent=1)
    for rel, t in trees.items():
        print(rel, "%d top-level keys" % len(t))
    print(OUT, "%.1f KB" % (os.path.getsize(OUT) / 1024))


if __name__ == "__main__":
    main()
