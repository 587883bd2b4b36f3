"""kernel-name normalisation shared by the profile tools: rocprofv3 leaves symbols whose template arguments contain `__bf16`
(Itanium `DF16b`) mangled, so `_ZN2cb16gemm_glds_kernelIDF16bLi128E...EEvNS_5KArgsE` becomes `gemm_glds_kernel<__bf16, 128, ...>`"""
import re


def _demangle_cb(sym):
    m = re.match(r"_ZN2cb(\d+)", sym)
    if not m:
        return sym
    n = int(m.group(1)); pos = m.end()
    name = sym[pos:pos + n]; pos += n
    if pos >= len(sym) or sym[pos] != "I":
        return name
    pos += 1
    args = []
    while pos < len(sym) and sym[pos] != "E":
        if sym.startswith("DF16b", pos):
            args.append("__bf16"); pos += 5
        elif sym[pos] == "f":
            args.append("float"); pos += 1
        elif sym[pos] == "L":
            e = sym.index("E", pos)
            lit = sym[pos + 1:e]
            if lit[0] == "b":
                args.append("true" if lit[1:] == "1" else "false")
            else:
                args.append(lit[1:].replace("n", "-"))
            pos = e + 1
        else:
            return name + "<?>"
    return f"{name}<{', '.join(args)}>"


def clean(kernel_name):
    n = re.sub(r"\(.*", "", kernel_name).replace("void cb::", "").replace("cb::", "").replace("(cb::_anonymous_namespace_)::", "")
    if n.startswith("_ZN2cb"):
        n = _demangle_cb(n)
    return n
