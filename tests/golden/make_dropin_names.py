"""AST-scan the reference's experiment scripts for every name they use on the ``diffusion_net`` package
(``diffusion_net.<module>.<name>`` attribute chains and ``from diffusion_net.<module> import <name>``) and write
tests/golden/dropin_names.json: {experiment dir: {"<module>.<name>": [file:line, ...]}}.  Dev container only (reads /root/reference).

    python tests/golden/make_dropin_names.py
"""
import ast
import json
import os
import sys

REF_EXP = "/root/reference/experiments"
HERE = os.path.dirname(os.path.abspath(__file__))


def scan_file(path):
    tree = ast.parse(open(path).read(), filename=path)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Attribute) and isinstance(node.value.value, ast.Name) \
                and node.value.value.id == "diffusion_net":
            found.setdefault("%s.%s" % (node.value.attr, node.attr), []).append(node.lineno)
        elif isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("diffusion_net"):
            mod = node.module[len("diffusion_net"):].lstrip(".")
            for a in node.names:
                found.setdefault(("%s.%s" % (mod, a.name)) if mod else a.name, []).append(node.lineno)
    return found


def scan(root=REF_EXP):
    out = {}
    for exp in sorted(os.listdir(root)):
        d = os.path.join(root, exp)
        if not os.path.isdir(d):
            continue
        names = {}
        for f in sorted(os.listdir(d)):
            if f.endswith(".py"):
                for k, lines in scan_file(os.path.join(d, f)).items():
                    names.setdefault(k, []).extend("%s:%d" % (f, ln) for ln in lines)
        out[exp] = {k: sorted(v) for k, v in sorted(names.items())}
    return out


if __name__ == "__main__":
    res = scan()
    json.dump(res, open(os.path.join(HERE, "dropin_names.json"), "w"), indent=1, sort_keys=True)
    for exp, names in res.items():
        print(exp, sorted(names))
