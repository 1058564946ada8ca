"""Static checks of the cgo shim (shim/) -- the image has no Go toolchain, so what `go build -tags hip` would reject is checked here.

CPU-only.  The checks that need the reference's sources (`/root/reference`, build container only) replay shim/manifest.json --
which files get `//go:build !hip`, which declarations move, which shim files are added -- and assert that the resulting package
is closed: every package-level identifier a kept file uses is declared by a kept file or by the shim, nothing
is declared twice, imports stay used, replaced functions keep the reference's signatures.  The checks against include/gnx_align.h
(every C symbol the shim binds exists, with the same number of arguments; every GNX_* constant and struct field exists) run anywhere.
"""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
MANIFEST = json.load(open(os.path.join(ROOT, "shim", "manifest.json")))
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "align")), reason="reference sources are only in the build container")

GO_KEYWORDS = set("break default func interface select case defer go map struct chan else goto package switch const fallthrough if range "
                  "type continue for import return var".split())


# ---------------------------------------------------------------------------------------------------------------------
# a small Go scanner: comments and literals blanked (newlines kept), top-level declarations with their spans
# ---------------------------------------------------------------------------------------------------------------------
def blank_comments_and_literals(src, keep_cgo_preamble=False):
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        two = src[i:i + 2]
        if two == "//":
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif two == "/*":
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j])); i = j
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('"' + " " * (j - i - 1) + '"'); i = j + 1
        elif c == "`":
            j = src.find("`", i + 1)
            out.append("`" + "".join(ch if ch == "\n" else " " for ch in src[i + 1:j]) + "`"); i = j + 1
        elif c == "'":
            j = i + 1
            while j < n and src[j] != "'":
                j += 2 if src[j] == "\\" else 1
            out.append("' '".ljust(j + 1 - i)[:j + 1 - i]); i = j + 1
        else:
            out.append(c); i += 1
    return "".join(out)


def match_close(text, i, open_ch, close_ch):
    depth = 0
    while i < len(text):
        if text[i] == open_ch:
            depth += 1
        elif text[i] == close_ch:
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise AssertionError("unbalanced %s" % open_ch)


class GoFile:
    def __init__(self, path, text=None):
        self.path = path
        self.name = os.path.basename(path)
        self.src = open(path).read() if text is None else text
        self.code = blank_comments_and_literals(self.src)
        self.decls = {}     # name -> (kind, start, end) of package-level declarations (methods excluded)
        self.imports = {}   # local name -> (path, start, end)
        self._scan()

    def _scan(self):
        code = self.code
        m = re.search(r"^package\s+(\w+)", code, re.M)
        self.package = m.group(1)
        depth, i, n = 0, 0, len(code)
        line_start = True
        while i < n:
            c = code[i]
            if depth == 0 and line_start:
                m = re.compile(r"(func|type|var|const|import)\b").match(code, i)
                if m:
                    i = self._decl(m.group(1), i)
                    line_start = False
                    continue
            if c in "{(":
                depth += 1
            elif c in "})":
                depth -= 1
            line_start = (c == "\n") or (line_start and c in " \t")
            i += 1

    def _decl(self, kind, i):
        code = self.code
        j = i + len(kind)
        while code[j] in " \t":
            j += 1
        if kind == "import":
            if code[j] == "(":
                end = match_close(code, j, "(", ")")
                body_start = j + 1
                for m in re.finditer(r'^[ \t]*(?:(\w+|\.|_)[ \t]+)?"', code[body_start:end], re.M):
                    q0 = body_start + m.end() - 1
                    q1 = code.index('"', q0 + 1)
                    self._add_import(m.group(1), self.src[q0 + 1:q1], body_start + m.start(), q1 + 1)
                return end + 1
            m = re.compile(r'(?:(\w+|\.|_)[ \t]+)?"').match(code, j)
            q0 = m.end() - 1
            q1 = code.index('"', q0 + 1)
            self._add_import(m.group(1), self.src[q0 + 1:q1], i, q1 + 1)
            return q1 + 1
        if kind == "func":
            if code[j] == "(":  # method: skip the receiver, do not record
                j = match_close(code, j, "(", ")") + 1
                name = None
            else:
                m = re.compile(r"\w+").match(code, j)
                name = m.group(0)
                j = m.end()
            # header ends at the first '{' at paren/bracket depth 0
            k, d = j, 0
            while not (code[k] == "{" and d == 0):
                d += code[k] in "(["
                d -= code[k] in ")]"
                k += 1
            end = match_close(code, k, "{", "}")
            if name:
                self.decls[name] = ("func", i, end + 1, code[j:k])
            return end + 1
        # type / var / const, single or grouped
        if code[j] == "(":
            end = match_close(code, j, "(", ")")
            d = 0
            pos = j + 1
            for line in code[j + 1:end].split("\n"):
                if d == 0:
                    m = re.match(r"[ \t]*(\w+(?:[ \t]*,[ \t]*\w+)*)", line)
                    if m:
                        for nm in re.split(r"[ \t]*,[ \t]*", m.group(1)):
                            self.decls[nm] = (kind, pos, pos + len(line), line)
                d += sum(line.count(ch) for ch in "{([") - sum(line.count(ch) for ch in "})]")
                pos += len(line) + 1
            return end + 1
        m = re.compile(r"\w+(?:[ \t]*,[ \t]*\w+)*").match(code, j)
        names = re.split(r"[ \t]*,[ \t]*", m.group(0))
        # the declaration ends at the end of the line on which all brackets are closed again
        k, d = m.end(), 0
        while k < len(code) and not (code[k] == "\n" and d == 0):
            d += code[k] in "{(["
            d -= code[k] in "})]"
            k += 1
        for nm in names:
            self.decls[nm] = (kind, i, k, code[m.end():k])
        return k

    def _add_import(self, alias, path, start, end):
        if path == "C":
            name = "C"
        else:
            name = alias if alias else path.rsplit("/", 1)[-1]
        self.imports[name] = (path, start, end)

    def without(self, decl_names, import_names):
        """the file after moving these declarations / imports elsewhere"""
        spans = [self.decls[d][1:3] for d in decl_names] + [self.imports[p][1:3] for p in import_names]
        text = self.src
        for a, b in sorted(spans, reverse=True):
            text = text[:a] + "".join(ch if ch == "\n" else " " for ch in text[a:b]) + text[b:]
        return GoFile(self.path, text)

    def identifiers(self):
        """identifier -> used as a plain name (not as .selector, not as a struct-literal / field key `name:`)"""
        used = set()
        for m in re.finditer(r"(?<![\w.])([A-Za-z_]\w*)", self.code):
            nm = m.group(1)
            if nm in GO_KEYWORDS:
                continue
            used.add(nm)
        return used

    def selectors(self):
        """package-qualified uses: {pkg: {Name, ...}}"""
        out = {}
        for m in re.finditer(r"(?<![\w.])([A-Za-z_]\w*)\.([A-Za-z_]\w*)", self.code):
            out.setdefault(m.group(1), set()).add(m.group(2))
        return out

    def local_names(self):
        """names bound inside the file by :=, var, parameters or results (over-approximation, used only to excuse a clash)"""
        names = set()
        for m in re.finditer(r"([\w, \t]+?):=", self.code):
            names.update(re.findall(r"[A-Za-z_]\w*", m.group(1)))
        for m in re.finditer(r"\bvar\s+([\w, \t]+?)\s+[\w\[\]*.]+", self.code):
            names.update(re.findall(r"[A-Za-z_]\w*", m.group(1)))
        for d in self.decls.values():
            if d[0] == "func":
                for m in re.finditer(r"([A-Za-z_]\w*)(?:\s*,\s*[A-Za-z_]\w*)*\s+(?:\[\]|\*|chan|<-|map|func|[A-Za-z_])", d[3]):
                    names.update(re.findall(r"[A-Za-z_]\w*", m.group(0).rsplit(None, 1)[0]))
        return names


def norm_sig(header):
    return re.sub(r"\s+", "", header)


def load_package(pkg, step):
    spec = MANIFEST["packages"][pkg]["steps"][step]
    ref_dir = os.path.join(REF, pkg)
    ref_files = [GoFile(os.path.join(ref_dir, f)) for f in sorted(os.listdir(ref_dir)) if f.endswith(".go")]
    ref_files = [f for f in ref_files if f.package == pkg]  # drops external test packages (package x_test), if any
    kept = []
    for f in ref_files:
        if f.name in spec["exclude_files"]:
            continue
        mv = spec["move_decls"].get(f.name)
        kept.append(f.without(mv["decls"], mv["imports"]) if mv else f)
    shim = [GoFile(os.path.join(ROOT, "shim", pkg, s)) for s in spec["shim_files"]]
    return spec, ref_files, kept, shim


STEPS = [(p, s) for p in MANIFEST["packages"] for s in MANIFEST["packages"][p]["steps"]]


@needs_ref
@pytest.mark.parametrize("pkg,step", STEPS)
def test_package_is_closed_under_the_hip_tag(pkg, step):
    spec, ref_files, kept, shim = load_package(pkg, step)
    old = {}
    for f in ref_files:
        for nm in f.decls:
            old.setdefault(nm, f.name)
    new = {}
    dup = []
    for f in kept + shim:
        for nm in f.decls:
            if nm in new and nm not in ("init", "_"):
                dup.append((nm, new[nm], f.name))
            new[nm] = f.name
    assert not dup, "declared twice under -tags hip: %r" % dup
    removed = set(old) - {nm for f in kept for nm in f.decls}
    assert removed, "the recipe removes nothing?"
    lost = set(old) - set(new)   # removed and not re-declared by the shim: nothing may refer to these any more
    problems = []
    for f in kept + shim:
        hits = (f.identifiers() & lost) - set(f.imports)
        hits -= f.local_names()  # a local variable that happens to share the name of a removed function
        for nm in sorted(hits):
            problems.append("%s uses %s (declared in %s, which the recipe removes)" % (f.name, nm, old[nm]))
    assert not problems, "\n".join(problems)
    # the moved declarations really exist where the manifest says, and the kept remainder of those files still uses its imports
    for fname, mv in spec["move_decls"].items():
        orig = next(f for f in ref_files if f.name == fname)
        for d in mv["decls"]:
            assert d in orig.decls, "%s does not declare %s" % (fname, d)
        for imp in mv["imports"]:
            assert imp in orig.imports
    for f in kept + shim:
        sel = f.selectors()
        for name, (path, _, _) in f.imports.items():
            if name in ("_", "."):
                continue
            assert name in sel, "%s: import %r is not used under -tags hip" % (f.name, path)
    # nothing the shim takes from the package is missing (lower-case helpers, types)
    for f in shim:
        for nm in sorted(f.identifiers() & set(old)):
            assert nm in new, "%s uses %s, which the recipe removes and nothing re-declares" % (f.name, nm)


@needs_ref
@pytest.mark.parametrize("pkg", list(MANIFEST["packages"]))
def test_replaced_functions_keep_their_signatures(pkg):
    spec = MANIFEST["packages"][pkg]
    ref_dir = os.path.join(REF, pkg)
    ref = {}
    for fn in sorted(os.listdir(ref_dir)):
        if fn.endswith(".go") and not fn.endswith("_test.go"):
            ref.update(GoFile(os.path.join(ref_dir, fn)).decls)
    shim = {}
    for fn in sorted(os.listdir(os.path.join(ROOT, "shim", pkg))):
        if fn.endswith(".go"):
            shim.update(GoFile(os.path.join(ROOT, "shim", pkg, fn)).decls)
    for nm in spec["same_signature"]:
        assert nm in shim, "shim does not declare %s" % nm
        assert ref[nm][0] == shim[nm][0] == "func"
        assert norm_sig(ref[nm][3]) == norm_sig(shim[nm][3]), "%s: signature differs\n ref : %s\n shim: %s" % (nm, ref[nm][3], shim[nm][3])
    for nm in spec["same_type"]:
        rf = next(GoFile(os.path.join(ref_dir, fn)) for fn in sorted(os.listdir(ref_dir)) if fn.endswith(".go") and nm in GoFile(os.path.join(ref_dir, fn)).decls)
        sf = next(GoFile(os.path.join(ROOT, "shim", pkg, fn)) for fn in sorted(os.listdir(os.path.join(ROOT, "shim", pkg))) if nm in GoFile(os.path.join(ROOT, "shim", pkg, fn)).decls)

        def body(f):
            _, a, _, _ = f.decls[nm]
            k = f.code.index("{", a)
            return norm_sig(f.code[k:match_close(f.code, k, "{", "}") + 1])
        assert body(rf) == body(sf), "type %s differs" % nm


# ---------------------------------------------------------------------------------------------------------------------
# shim <-> include/gnx_align.h (runs anywhere)
# ---------------------------------------------------------------------------------------------------------------------
def header_model():
    h = open(os.path.join(ROOT, "include", "gnx_align.h")).read()
    code = blank_comments_and_literals(h)
    protos = {}
    for m in re.finditer(r"\b(gnx_\w+)\s*\(", code):
        end = match_close(code, m.end() - 1, "(", ")")
        if code[end + 1:end + 3].lstrip().startswith(";"):
            args = code[m.end():end].strip()
            protos[m.group(1)] = 0 if args in ("", "void") else len(split_top(args))
    consts = set(re.findall(r"#define\s+(GNX_\w+)", code)) | set(re.findall(r"\b(GNX_[A-Z0-9_]+)\s*=", code))
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{", code):
        end = match_close(code, m.end() - 1, "{", "}")
        fields = set()
        for decl in code[m.end():end].split(";"):
            for part in decl.split(","):  # `int64_t a, b, c`
                mm = re.search(r"(\w+)\s*(?:\[[^\]]*\])?\s*$", part.strip())
                if mm:
                    fields.add(mm.group(1))
        structs[m.group(1)] = fields
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", code):  # opaque handles
        structs.setdefault(m.group(2), set())
    return protos, consts, structs


def split_top(s):
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur)); cur = []
        else:
            cur.append(ch)
    parts.append("".join(cur))
    return [p for p in parts if p.strip()]


def shim_files():
    out = []
    for pkg in MANIFEST["packages"]:
        d = os.path.join(ROOT, "shim", pkg)
        out += [GoFile(os.path.join(d, f)) for f in sorted(os.listdir(d)) if f.endswith(".go")]
    return out


def test_every_shim_file_is_in_the_manifest_and_tagged():
    for pkg, spec in MANIFEST["packages"].items():
        listed = set()
        for st in spec["steps"].values():
            listed.update(st["shim_files"])
        on_disk = {f for f in os.listdir(os.path.join(ROOT, "shim", pkg)) if f.endswith(".go")}
        assert listed == on_disk
        for f in on_disk:
            src = open(os.path.join(ROOT, "shim", pkg, f)).read()
            assert re.search(r"^//go:build hip$", src, re.M), f
            assert re.search(r"^package %s$" % pkg, src, re.M), f
            assert re.search(r"^//go:build hip$", src, re.M).start() < re.search(r"^package %s$" % pkg, src, re.M).start()
            assert 'import "C"' in src and "gnx_align.h" in src


def test_shim_binds_only_what_the_header_declares():
    protos, consts, structs = header_model()
    assert "gnx_align_batch" in protos and protos["gnx_align_pair"] == 8
    bound = set()
    for f in shim_files():
        for m in re.finditer(r"\bC\.(\w+)", f.code):
            nm = m.group(1)
            if nm.startswith("gnx_") and f.code[m.end():m.end() + 1] == "(":
                assert nm in protos, "%s calls C.%s, not declared in gnx_align.h" % (f.name, nm)
                end = match_close(f.code, m.end(), "(", ")")
                nargs = len(split_top(f.code[m.end() + 1:end]))
                assert nargs == protos[nm], "%s: C.%s called with %d arguments, the header declares %d" % (f.name, nm, nargs, protos[nm])
                bound.add(nm)
            elif nm.startswith("GNX_"):
                assert nm in consts, "%s uses C.%s, not defined in gnx_align.h" % (f.name, nm)
            elif nm.startswith("gnx_"):
                assert nm in structs, "%s uses type C.%s, not declared in gnx_align.h" % (f.name, nm)
    # what SURVEY 8b lists for the boundary is bound by the shim
    for need in ("gnx_align_pair", "gnx_align_batch", "gnx_align_batch_by_offset", "gnx_set_reference", "gnx_init_devices", "gnx_free",
                 "gnx_last_error", "gnx_affine_gap_chunk_batch", "gnx_multiple_affine_gap_batch", "gnx_gsw_extend_batch", "gnx_get_timing",
                 "gnx_gsw_graph_create", "gnx_gsw_map_reads", "gnx_gsw_graph_free"):
        assert need in bound, need
    # struct fields the shim touches exist
    for f in shim_files():
        for var, typ in re.findall(r"\bvar\s+(\w+)\s+C\.(gnx_\w+)", f.code):
            for fld in re.findall(r"\b%s\.(\w+)" % re.escape(var), f.code):
                assert fld in structs[typ], "%s: %s.%s is not a field of %s" % (f.name, var, fld, typ)
        for m in re.finditer(r"\b(\w+)\.(run_length|op)\b", f.code):
            assert m.group(2) in structs["gnx_cigar"]


def test_shim_never_indexes_an_empty_batch():
    """`&x[0]` of a possibly empty slice panics in Go; every batch function returns early for n == 0."""
    for f in shim_files():
        for nm, d in f.decls.items():
            if d[0] != "func":
                continue
            body = f.code[d[1]:d[2]]
            if re.search(r"&\w+\[0\]", body) and re.search(r"\bn\s*:=\s*len\(", body):
                assert re.search(r"if\s+n\s*==\s*0\s*\{\s*return", body), "%s.%s indexes [0] without an n == 0 guard" % (f.name, nm)
