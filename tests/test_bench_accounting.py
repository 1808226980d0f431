"""The FLOP / byte accounting that bench.py's roofline object is computed from (CPU only)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_posenet_forward_flops_match_survey():
    # posenet3d_50 forward = 995.4 GFLOP per 128^3 sample (PyTorch FlopCounterMode, 2 x MAC)
    f = bench.posenet_conv_flops(128, 128, 1)
    fwd = sum(f[k] for k in ("conv_igemm_stem", "conv_igemm_k1", "conv_igemm_k3", "conv_igemm_deconv"))
    assert abs(fwd / 1e9 - 995.4) < 0.1
    # every convolution has a weight gradient of the same cost; every one but none is skipped for data gradients
    assert f["conv_wgrad"] == fwd
    assert f["conv_igemm_dgrad"] + f["conv_stem_dgrad"] == fwd
    # 128x128x512 is exactly 4x the voxels
    g = bench.posenet_conv_flops(512, 128, 1)
    assert all(abs(g[k] / f[k] - 4.0) < 1e-12 for k in f)
    assert abs(sum(g[k] for k in ("conv_igemm_stem", "conv_igemm_k1", "conv_igemm_k3", "conv_igemm_deconv")) / 1e9 - 3981.4) < 0.5


def test_lct_algorithmic_bytes():
    V = 128 ** 3
    total = sum(bench.algorithmic_work(k, 128, 128, 2)[1] for k in
                ("lct_axis_fwd_t", "lct_axis_fwd_h", "lct_axis_mid_w", "lct_axis_inv_h", "lct_axis_inv_t"))
    # one packed pair (2 samples): (8V + 16V) + 48V + (64V + 64V) + 48V + (16V + 8V) = 272V bytes = 136V per sample
    assert total == 272 * V


def test_bench_functions_reference_only_defined_names():
    """bench.py's secondary workloads only run on a GPU box; catch NameErrors (a stale variable after an edit)
    here: every name a function loads must be a local, an enclosing/global binding, or a builtin."""
    import ast
    import builtins
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    tree = ast.parse(open(path).read())
    module_names = set(dir(builtins))
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            module_names.update((a.asname or a.name).split(".")[0] for a in node.names)
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            module_names.add(node.name)
        elif isinstance(node, ast.Assign):
            for t in node.targets:
                module_names.update(n.id for n in ast.walk(t) if isinstance(n, ast.Name))

    def check(fn, outer):
        bound = set(outer)
        bound.update(a.arg for a in fn.args.args + fn.args.kwonlyargs)
        if fn.args.vararg:
            bound.add(fn.args.vararg.arg)
        if fn.args.kwarg:
            bound.add(fn.args.kwarg.arg)
        for n in ast.walk(fn):
            if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                bound.add(n.id)
            elif isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n is not fn:
                bound.add(n.name)
                if isinstance(n, ast.FunctionDef):  # nested helper: its parameters (coarse: one shared scope)
                    bound.update(a.arg for a in n.args.args + n.args.kwonlyargs)
            elif isinstance(n, (ast.Import, ast.ImportFrom)):
                bound.update((a.asname or a.name).split(".")[0] for a in n.names)
            elif isinstance(n, ast.ExceptHandler) and n.name:
                bound.add(n.name)
            elif isinstance(n, (ast.Lambda,)):
                bound.update(a.arg for a in n.args.args)
            elif isinstance(n, ast.comprehension):
                bound.update(x.id for x in ast.walk(n.target) if isinstance(x, ast.Name))
        missing = sorted({n.id for n in ast.walk(fn) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)} - bound)
        assert not missing, f"{fn.name}: undefined names {missing}"

    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            check(node, module_names)
