"""Tiny static check (no pyflakes in this image): names loaded in a function that are bound nowhere -- not in the function,
an enclosing function, the module or builtins.  Conservative: flags only what would raise NameError."""
import ast
import builtins
import sys


def bound_names(node):
    out = set()
    for n in ast.walk(node):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
        elif isinstance(n, ast.arg):
            out.add(n.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                out.add((a.asname or a.name).split('.')[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            out.update(n.names)
    return out


def check(path):
    tree = ast.parse(open(path).read(), path)
    known = bound_names(tree) | set(dir(builtins)) | {'__file__', '__name__', '__doc__'}
    bad = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in known:
            bad.append((n.lineno, n.id))
    return bad


if __name__ == '__main__':
    rc = 0
    for p in sys.argv[1:]:
        for line, name in check(p):
            print(f'{p}:{line}: undefined name {name!r}')
            rc = 1
    sys.exit(rc)
