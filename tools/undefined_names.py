"""Poor man's pyflakes (none in the image): names a module loads that nothing in it binds.  usage: undefined_names.py FILE..."""
import ast
import builtins
import sys


def bound_names(node):
    out = set()
    for n in ast.walk(node):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            a = n.args
            for x in a.args + a.kwonlyargs + a.posonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                out.add(x.arg)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                out.add((al.asname or al.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
    return out


def main():
    rc = 0
    for path in sys.argv[1:]:
        tree = ast.parse(open(path).read(), path)
        known = bound_names(tree) | set(dir(builtins)) | {"__file__", "__name__"}
        for n in ast.walk(tree):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in known:
                print("%s:%d: undefined name %r" % (path, n.lineno, n.id))
                rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
