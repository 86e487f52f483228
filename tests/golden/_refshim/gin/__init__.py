"""Minimal stand-in for gin-config, used ONLY by tests/golden/make_golden.py in
the build container to import the reference's Python (which decorates its
classes with @gin.configurable).  No gin files are parsed: the generator passes
the constructor kwargs of SURVEY.md Appendix A explicitly."""


def configurable(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(obj):
        return obj

    return deco


def add_config_file_search_path(*a, **k):
    return None


def parse_config_file(*a, **k):
    raise RuntimeError("gin shim: config parsing is not available")


def bind_parameter(*a, **k):
    return None


def operative_config_str():
    return ""


class _ConstantsShim:
    pass


REQUIRED = object()
