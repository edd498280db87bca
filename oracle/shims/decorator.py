"""The caller protocol of the `decorator` package (>= 4.2) that detection/utils/ops.py:408-433 relies on:

    @decorator
    def caller(func, opt=default, *args, **kw): ...

makes `caller` usable bare (`@caller`) and as a factory (`@caller(opt=value)`); the decorated function is called as
`caller(func, *(opts + args), **kw)` with the opts taken from the factory call or the caller's defaults."""
import functools
import inspect


def decorator(caller):
    sig = inspect.signature(caller)
    params = list(sig.parameters.values())[1:]
    opts = [p for p in params if p.kind == p.POSITIONAL_OR_KEYWORD and p.default is not p.empty]

    def apply(func, values):
        @functools.wraps(func)
        def run(*args, **kw):
            return caller(func, *(tuple(values) + args), **kw)
        return run

    def dec(func=None, *args, **kw):
        values = [kw.get(p.name, args[i] if i < len(args) else p.default) for i, p in enumerate(opts)]
        if func is None or not callable(func):
            return lambda f: apply(f, values)
        return apply(func, values)

    return functools.wraps(caller)(dec)
