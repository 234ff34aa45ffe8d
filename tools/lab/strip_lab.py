"""Resolve the lab conditionals of a kernel source to their shipped branch: every #if / #ifdef / #ifndef whose condition
names only GEOT_*LAB* macros is evaluated with those macros UNDEFINED, and the tunables they default
(#ifndef X / #define X v / #endif) become plain constants at their uses.  The lab copies with the knock-outs intact stay
under tools/lab/kernels/ (build one with: hipcc ... -D<macro> tools/lab/kernels/<file>.hip).
usage: python tools/lab/strip_lab.py geot_amd/csrc/<file>.hip ..."""
import re
import sys

LAB = re.compile(r"\bGEOT_(?:[A-Z0-9]+_)?LAB_[A-Z0-9_]+\b")


def cond_value(line):
    """None if the directive is not a pure lab condition, else its truth value with every lab macro undefined."""
    m = re.match(r"\s*#\s*(ifdef|ifndef)\s+(\w+)\s*(?://.*)?$", line)
    if m:
        if not LAB.fullmatch(m.group(2)):
            return None
        return m.group(1) == "ifndef"
    m = re.match(r"\s*#\s*(if|elif)\s+(.*?)\s*(?://.*)?$", line)
    if m:
        expr = m.group(2)
        names = set(re.findall(r"\b[A-Za-z_]\w*\b", expr)) - {"defined"}
        if not names or not all(LAB.fullmatch(n) for n in names):
            return None
        py = re.sub(r"defined\s*\(\s*\w+\s*\)", "False", expr)
        py = re.sub(r"defined\s+\w+", "False", py)
        py = LAB.sub("0", py).replace("||", " or ").replace("&&", " and ").replace("!", " not ")
        return bool(eval(py))
    return None


def strip(text):
    out, stack = [], []      # stack entries: None (foreign conditional) or dict(taken=bool, done=bool)
    for line in text.split("\n"):
        s = line.strip()
        active = all(e is None or e["taken"] for e in stack)
        if re.match(r"#\s*(if|ifdef|ifndef)\b", s):
            v = cond_value(line)
            if v is None:
                stack.append(None)
                if active:
                    out.append(line)
            else:
                stack.append({"taken": v, "done": v})
            continue
        if re.match(r"#\s*elif\b", s) and stack and stack[-1] is not None:
            e = stack[-1]
            v = cond_value(line)
            e["taken"] = (not e["done"]) and bool(v)
            e["done"] = e["done"] or e["taken"]
            continue
        if re.match(r"#\s*else\b", s) and stack and stack[-1] is not None:
            e = stack[-1]
            e["taken"] = not e["done"]
            e["done"] = True
            continue
        if re.match(r"#\s*endif\b", s):
            e = stack.pop()
            if e is None and all(x is None or x["taken"] for x in stack):
                out.append(line)
            continue
        if active:
            out.append(line)
    text = "\n".join(out)
    # tunables: '#define GEOT_X_LAB_Y value' left by the resolved #ifndef -> constants at the uses
    for m in list(re.finditer(r"^#define\s+(GEOT_(?:[A-Z0-9]+_)?LAB_[A-Z0-9_]+)\s+(\S+)\s*$", text, re.M)):
        name, val = m.group(1), m.group(2)
        text = text.replace(m.group(0) + "\n", "")
        text = re.sub(r"\b%s\b" % name, val, text)
    return text


if __name__ == "__main__":
    for path in sys.argv[1:]:
        src = open(path).read()
        new = strip(src)
        left = LAB.findall(new)
        open(path, "w").write(new)
        print("%s: %d -> %d lines, lab names left: %s" % (path, src.count("\n"), new.count("\n"), sorted(set(left))))
