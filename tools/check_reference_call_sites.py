#!/usr/bin/env python3
"""Build-container-only check of the drop-in boundary: reads the EIGHT call sites
`MDCONV_CUDA.<name>(...)` of the reference's own Python wrapper with `ast` (nothing of the reference is copied,
committed or shipped -- the file is parsed where it lies) and asserts that the function of the same name in
`modulated_deform_conv_amd/MDCONV_CUDA.py` takes the same number of positional arguments IN THE SAME ORDER.

The order is compared through a role name derived from each argument expression of the call site:
    input, weight, ...                      ->  the variable's name         (Name)
    weight.shape[2 + i]                     ->  kernel<i>
    ctx.stride[i] / padding / dilation      ->  stride<i> / pad<i> / dilation<i>
    ctx.groups / deformable_groups / ...    ->  group / deformable_group / in_step / with_bias
and through the parameter names of our binding mapped the same way (kernel_h -> kernel0, pad_w -> pad1, ...).
Also checks how many values the call site unpacks (the 5-tuple of the MDCN2d backward, one tensor of its forward).

    python tools/check_reference_call_sites.py [/root/reference]        exit 0 = all eight agree
`tests/test_abi_cpu.py::test_extension_module_surface_matches_reference` runs it when /root/reference exists
(this container) and falls back to the committed arity table elsewhere (the GPU box has no reference tree)."""
import ast
import inspect
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AXES = {"h": 0, "w": 1, "l": 2}
CTX = {"stride": "stride", "padding": "pad", "dilation": "dilation"}
CTX_SCALAR = {"groups": "group", "deformable_groups": "deformable_group", "in_step": "in_step", "with_bias": "with_bias"}


def role_of_call_arg(node):
    """Role name of one argument expression of a reference call site."""
    if isinstance(node, ast.Name):
        return node.id
    if isinstance(node, ast.Subscript):
        idx = node.slice
        if isinstance(idx, ast.Index):          # python < 3.9
            idx = idx.value
        i = idx.value if isinstance(idx, ast.Constant) else None
        base = node.value
        if isinstance(base, ast.Attribute) and isinstance(base.value, ast.Name):
            if base.value.id == "weight" and base.attr == "shape" and i is not None:
                return "kernel%d" % (i - 2)
            if base.value.id == "ctx" and base.attr in CTX and i is not None:
                return "%s%d" % (CTX[base.attr], i)
    if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "ctx":
        if node.attr in CTX_SCALAR:
            return CTX_SCALAR[node.attr]
    raise ValueError("unrecognised argument expression: %s" % ast.dump(node))


def role_of_param(name):
    """Role name of one parameter of our binding."""
    for prefix in ("kernel", "stride", "pad", "dilation"):
        if name.startswith(prefix + "_") and name[len(prefix) + 1:] in AXES:
            return "%s%d" % (prefix, AXES[name[len(prefix) + 1:]])
    return name


def reference_call_sites(path):
    """-> {function name: (line, [roles], number of values the statement unpacks or None)}"""
    tree = ast.parse(open(path).read(), filename=path)
    sites = {}
    for node in ast.walk(tree):
        call, targets = None, None
        if isinstance(node, ast.Expr) and isinstance(node.value, ast.Call):
            call = node.value
        elif isinstance(node, ast.Assign) and isinstance(node.value, ast.Call):
            call = node.value
            t = node.targets[0]
            targets = len(t.elts) if isinstance(t, (ast.Tuple, ast.List)) else 1
        if call is None:
            continue
        f = call.func
        if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id == "MDCONV_CUDA":
            assert not call.keywords, "keyword arguments at %s:%d" % (path, call.lineno)
            assert f.attr not in sites, "two call sites of %s" % f.attr
            sites[f.attr] = (call.lineno, [role_of_call_arg(a) for a in call.args], targets)
    return sites


def check(reference_root="/root/reference", verbose=False):
    path = os.path.join(reference_root, "modulated_deform_conv.py")
    sites = reference_call_sites(path)
    assert len(sites) == 8, "expected the 8 call sites of SURVEY.md section 8b, found %d: %s" % (len(sites), sorted(sites))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from modulated_deform_conv_amd import MDCONV_CUDA as M
    arity = {}
    for name, (line, roles, targets) in sorted(sites.items(), key=lambda kv: kv[1][0]):
        fn = getattr(M, name, None)
        assert fn is not None, "MDCONV_CUDA.%s (reference call site :%d) is not exported" % (name, line)
        ours = [role_of_param(p) for p in inspect.signature(fn).parameters]
        assert ours == roles, "%s: positional order differs from modulated_deform_conv.py:%d\n  reference: %s\n  ours:      %s" % (
            name, line, roles, ours)
        arity[name] = len(roles)
        if verbose:
            print("%-42s :%-4d %2d positional arguments, order identical%s" % (
                name, line, len(roles), "" if targets is None else ", %d value(s) unpacked" % targets))
    # return-value shapes of the two entry points the reference takes results from (modulated_deform_conv.py:112, 142)
    assert sites["modulated_deform_conv2d_forward_cuda"][2] == 1
    assert sites["modulated_deform_conv2d_backward_cuda"][2] == 5
    for name in sites:
        if name not in ("modulated_deform_conv2d_forward_cuda", "modulated_deform_conv2d_backward_cuda"):
            assert sites[name][2] is None, "%s: the reference ignores its return value" % name
    return arity


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    if not os.path.isdir(root):
        sys.exit("no reference tree at %s (this check runs in the build container only)" % root)
    check(root, verbose=True)
    print("all eight call sites agree with modulated_deform_conv_amd/MDCONV_CUDA.py")
