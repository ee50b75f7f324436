#!/usr/bin/env python3
"""A compiler is not available for bindings/rust/*.rs (no rustc in the build image), so this lint checks the one thing
that sank the first version of the shim: names that do not exist in the reference.  For every shim file

  * every `crate::a::b::C` path (in `use` items and inline) must resolve in the Fyrox tree: the module file exists and
    defines or re-exports the last segment;
  * every `.method(` called on a value and every `Type::function(` must exist as a `pub fn` somewhere in the reference
    crates the path touches (or be a std / nalgebra / shim-own name from the allow-lists below);
  * every `Enum::Variant` used in a pattern must be a variant of that reference enum;
  * a method that only a TRAIT of the reference declares (e.g. `SceneGraph::try_get_node`) needs that trait in a `use` of the file;
  * every extern function called (`fyx_*`) must be declared in bindings/rust/fyrox_hip_sys.rs (generated from the header)
    and be called with as many arguments as it is declared with.

It is a grep-level check, not a type checker: it cannot see a wrong receiver type or a borrow error.  It reads
/root/reference, so it only runs where the reference is (the build container), and the CPU test that calls it skips
elsewhere.   python tools/lint_rust_shim.py  ->  exit code 0 / 1, findings on stdout."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SHIM = [os.path.join(ROOT, "bindings", "rust", f) for f in ("fyrox_hip.rs", "fyrox_hip_flatten.rs")]
SYS = os.path.join(ROOT, "bindings", "rust", "fyrox_hip_sys.rs")
CRATES = ["fyrox-impl/src", "fyrox-animation/src", "fyrox-core/src", "fyrox-math/src", "fyrox-resource/src", "fyrox-graph/src"]

# methods of std, core, nalgebra, fxhash and the shim's own types (not defined in the reference tree)
STD_METHODS = set("""
len iter iter_mut enumerate map filter collect copied cloned unwrap_or unwrap_or_else unwrap_or_default map_err ok_or
is_none is_some insert get get_mut entry or_insert push extend_from_slice as_ptr as_mut_ptr as_slice as_ref as_mut to_string
to_string_lossy into_owned into clone then copy_from_slice is_empty with_capacity flat_map and_then contains default
to_rotation_matrix matrix try_inverse identity new new_unchecked from_ptr null null_mut keys values to_bits finish hash ok resize_with push_back pop_front unwrap_or_default
""".split())
SHIM_OWN = set("raw id check node from_parts upload_tracks create_rig create_bone_list from_player attach_machine sync_parameters rebuild_machine reset_layer sync_machine push_animations pull_animations of pop_layer_event pop_layer_event_raw".split())


def count_args(src: str, open_paren: int):
    """arguments of the call whose `(` is at src[open_paren]; None when the brackets do not balance"""
    depth, n, seen = 0, 0, False
    for i in range(open_paren, len(src)):
        ch = src[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                return n + (1 if seen else 0)
        elif depth == 1:
            if ch == ",":
                n += 1
                seen = False
            elif not ch.isspace():
                seen = True
        if depth > 1 and not ch.isspace():
            seen = True
    return None


def ref_sources():
    out = {}
    for c in CRATES:
        for dp, _, files in os.walk(os.path.join(REF, c)):
            for f in files:
                if f.endswith(".rs"):
                    p = os.path.join(dp, f)
                    out[p] = open(p, errors="replace").read()
    return out


def strip_comments_and_strings(src):
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r'"(?:\\.|[^"\\])*"', '""', src)
    return src


def module_file(crate_dir, segments):
    """the file that holds module crate::seg0::seg1::...: .../seg.rs or .../seg/mod.rs; None if a segment is not a module"""
    d = os.path.join(REF, crate_dir)
    path = os.path.join(d, "lib.rs")
    for s in segments:
        cand = [os.path.join(d, s + ".rs"), os.path.join(d, s, "mod.rs")]
        hit = next((c for c in cand if os.path.exists(c)), None)
        if hit is None:
            return path, False
        path = hit
        d = os.path.join(d, s) if hit.endswith("mod.rs") else os.path.join(d, s)
    return path, True


# crate-level re-exports the shim relies on (fyrox-impl/src/lib.rs:50-60): crate::core = fyrox_core, crate::generic_animation = fyrox_animation
ALIASES = {"core": "fyrox-core/src", "generic_animation": "fyrox-animation/src", "graph": "fyrox-graph/src"}


def check_path(path, sources, findings, where):
    segs = path.split("::")
    if segs[0] != "crate" or len(segs) < 3:
        return
    crate_dir, rest = "fyrox-impl/src", segs[1:]
    if rest[0] in ALIASES:
        crate_dir, rest = ALIASES[rest[0]], rest[1:]
        if rest and rest[0] == "math" and crate_dir == "fyrox-core/src":      # fyrox_core::math = fyrox_math (fyrox-core/src/lib.rs)
            crate_dir, rest = "fyrox-math/src", rest[1:]
        if rest and rest[0] in ("algebra", "uuid"):       # nalgebra / uuid re-exports: outside the tree
            return
    if not rest:
        return
    item, mods = rest[-1], rest[:-1]
    f, all_modules = module_file(crate_dir, mods)
    src = sources.get(f) or (open(f, errors="replace").read() if os.path.exists(f) else "")
    if not all_modules:
        # the last "module" segment may itself be a type (Type::item): accept if the type is defined in the file reached
        pass
    pat = re.compile(r"\b(pub\s+)?(struct|enum|type|trait|fn|mod|const|static)\s+" + re.escape(item) + r"\b|pub\s+use\s+[^;]*\b" + re.escape(item) + r"\b", re.S)
    found = bool(pat.search(src))
    if not found:      # one level of glob re-exports: `pub use handle::*;`
        for g in re.findall(r"pub\s+use\s+(?:self::)?([a-z_]+)::\*;", src):
            base = os.path.dirname(f)
            for cand in (os.path.join(base, g + ".rs"), os.path.join(base, g, "mod.rs")):
                if os.path.exists(cand) and pat.search(open(cand, errors="replace").read()):
                    found = True
    if not found:
        findings.append(f"{where}: path `{path}`: `{item}` is neither defined nor re-exported in {os.path.relpath(f, REF)}")


def main():
    if not os.path.isdir(REF):
        print("reference tree not present: nothing checked")
        return 0
    sources = ref_sources()
    all_ref = "\n".join(sources.values())
    pub_fns = set(re.findall(r"pub(?:\([a-z]+\))?\s+(?:const\s+)?(?:unsafe\s+)?fn\s+([a-zA-Z_0-9]+)", all_ref))
    trait_fns = set(re.findall(r"^\s+fn\s+([a-zA-Z_0-9]+)", all_ref, flags=re.M))       # trait methods are not `pub fn`
    pub_fields = set(re.findall(r"^\s+pub\s+([a-z_0-9]+)\s*:", all_ref, flags=re.M))
    # methods DECLARED by a trait of the reference: callable only with that trait in scope (`use ...::Trait;`)
    trait_decl = {}
    for m in re.finditer(r"pub\s+trait\s+([A-Za-z0-9_]+)[^{;]*\{", all_ref):
        depth, i = 1, m.end()
        while depth and i < len(all_ref):
            depth += all_ref[i] == "{"
            depth -= all_ref[i] == "}"
            i += 1
        for fn in re.findall(r"^\s+fn\s+([a-zA-Z_0-9]+)", all_ref[m.end():i], flags=re.M):
            trait_decl.setdefault(fn, set()).add(m.group(1))
    enums = {}
    for m in re.finditer(r"pub\s+enum\s+([A-Za-z0-9_]+)[^{]*\{", all_ref):
        body, depth, i = "", 1, m.end()
        while depth and i < len(all_ref):
            ch = all_ref[i]
            depth += ch == "{"
            depth -= ch == "}"
            body += ch
            i += 1
        body = re.sub(r"//[^\n]*", "", body)
        enums.setdefault(m.group(1), set()).update(re.findall(r"^\s{4}([A-Z][A-Za-z0-9_]*)", body, flags=re.M))
    sys_fns = set(re.findall(r"pub fn (fyx_[a-z0-9_]+)", open(SYS).read()))
    sys_arity = {}
    for m in re.finditer(r"pub fn (fyx_[a-z0-9_]+)\(([^)]*)\)", open(SYS).read()):
        args = [a for a in m.group(2).split(",") if a.strip()]
        sys_arity[m.group(1)] = len(args)
    sys_consts = set(re.findall(r"pub const (FYX_[A-Z0-9_]+)", open(SYS).read()))
    sys_structs = set(re.findall(r"pub struct (Fyx[A-Za-z0-9]+)", open(SYS).read()))
    findings = []
    for shim in SHIM:
        name = os.path.basename(shim)
        raw = open(shim).read()
        src = strip_comments_and_strings(raw)
        # ---- paths: `use crate::{a::{B, C}, d::E};` trees are expanded, then inline paths
        def expand(prefix, tree):
            tree = tree.strip()
            if tree.startswith("{"):
                depth, cur, parts = 0, "", []
                for ch in tree[1:-1]:
                    if ch == "," and depth == 0:
                        parts.append(cur); cur = ""
                    else:
                        depth += ch == "{"
                        depth -= ch == "}"
                        cur += ch
                if cur.strip():
                    parts.append(cur)
                for p_ in parts:
                    yield from expand(prefix, p_)
            elif "::{" in tree:
                head, rest = tree.split("::{", 1)
                yield from expand(prefix + "::" + head.strip() if prefix else head.strip(), "{" + rest)
            else:
                t = tree.strip()
                if t and t != "self" and t != "*":
                    yield (prefix + "::" + t) if prefix else t
        for m in re.finditer(r"\buse\s+(crate::[^;]+);", src):
            for pth in expand("", m.group(1).replace("\n", " ")):
                check_path(re.sub(r"\s+", "", pth), sources, findings, name)
        for m in re.finditer(r"\bcrate(?:::[A-Za-z_][A-Za-z0-9_]*)+", re.sub(r"\buse\s+[^;]+;", "", src)):
            check_path(m.group(0), sources, findings, name)
        # ---- methods
        for m in re.finditer(r"\.([a-z_][a-z0-9_]*)\s*\(", src):
            fn = m.group(1)
            if fn not in STD_METHODS and fn not in SHIM_OWN and fn not in pub_fns and fn in trait_decl:
                used = set(re.findall(r"\b([A-Z][A-Za-z0-9_]*)\b", " ".join(re.findall(r"\buse\s+[^;]+;", src))))
                if not (trait_decl[fn] & used):
                    line = src[:m.start()].count("\n") + 1
                    findings.append(f"{name}:{line}: `.{fn}(` is a method of the trait(s) {sorted(trait_decl[fn])}: none of them is in a `use` of this file")
                continue
            if fn in STD_METHODS or fn in SHIM_OWN or fn in pub_fns or fn in trait_fns:
                continue
            line = src[:m.start()].count("\n") + 1
            findings.append(f"{name}:{line}: `.{fn}(` is not a pub fn of the reference (nor std / nalgebra / the shim)")
        # ---- field reads on reference values: `.field` not followed by `(`; only names that look like reference fields
        for m in re.finditer(r"\b[a-z_][a-z0-9_]*\.([a-z_][a-z0-9_]*)\b(?!\s*\()", src):
            fld = m.group(1)
            if fld in pub_fields or fld in ("x", "y", "z", "w", "coords", "0", "start", "end", "ctx", "hip", "id", "n_instances",
                                            "signal_names", "index_of", "rig_id", "animation_index", "parameter_index", "positions",
                                            "normals", "tangents", "aabb", "has", "delta_position", "delta_rotation", "present", "kind",
                                            "value", "len",
                                            # LayerMaps / SavedLayer of fyrox_hip_flatten.rs (the shim's own structs)
                                            "layers", "node_index", "state_index", "transition_index", "by_index_nodes",
                                            "active_state", "active_transition", "transitions", "by_index", "known_states", "signature",
                                            # AnimationShadow
                                            "speed", "looped", "enabled", "slice", "time", "pending_layer_events", "a", "b"):
                continue
            if fld in pub_fns or fld in trait_fns:        # a method reference passed as a value
                continue
            line = src[:m.start()].count("\n") + 1
            findings.append(f"{name}:{line}: `.{fld}` is not a pub field of any reference struct")
        # ---- enum variants in patterns / expressions
        for m in re.finditer(r"\b([A-Z][A-Za-z0-9]+)::([A-Z][A-Za-z0-9]+)\b", src):
            en, var = m.group(1), m.group(2)
            if en in ("HipError", "Self", "Vec", "FxHashMap", "Some", "Matrix3", "Vector3", "Vector2", "Vector4", "Quaternion",
                      "UnitQuaternion", "CStr", "HipAnimator", "HipSkinning", "RigMap", "String", "Ok", "Err"):
                continue
            if en in enums and var in enums[en]:
                continue
            if en in enums:
                line = src[:m.start()].count("\n") + 1
                findings.append(f"{name}:{line}: `{en}::{var}`: `{var}` is not a variant of the reference's enum {en} ({sorted(enums[en])})")
        # ---- brackets balance (the cheapest syntax check there is; char literals and lifetimes are not brackets)
        flat = re.sub(r"'(\\.|[^'\\])'", "' '", src)
        stack = []
        pairs = {")": "(", "]": "[", "}": "{"}
        for i, ch in enumerate(flat):
            if ch in "([{":
                stack.append((ch, i))
            elif ch in ")]}":
                if not stack or stack[-1][0] != pairs[ch]:
                    findings.append(f"{name}:{flat[:i].count(chr(10)) + 1}: unbalanced `{ch}`")
                    break
                stack.pop()
        else:
            if stack:
                findings.append(f"{name}:{flat[:stack[-1][1]].count(chr(10)) + 1}: `{stack[-1][0]}` is never closed")
        # ---- FFI
        for m in re.finditer(r"\b(fyx_[a-z0-9_]+)\s*\(", src):
            if m.group(1) not in sys_fns:
                findings.append(f"{name}: extern `{m.group(1)}` is not declared in fyrox_hip_sys.rs")
                continue
            # the number of arguments of the call against the declaration (top-level commas inside balanced brackets)
            n_call = count_args(src, m.end() - 1)
            if n_call is not None and n_call != sys_arity[m.group(1)]:
                line = src[:m.start()].count("\n") + 1
                findings.append(f"{name}:{line}: `{m.group(1)}` is called with {n_call} argument(s), declared with {sys_arity[m.group(1)]}")
        for m in re.finditer(r"\b(FYX_[A-Z0-9_]+)\b", src):
            if m.group(1) not in sys_consts:
                findings.append(f"{name}: constant `{m.group(1)}` is not declared in fyrox_hip_sys.rs")
        for m in re.finditer(r"\b(Fyx[A-Z][A-Za-z0-9]+)\b", src):
            if m.group(1) not in sys_structs:
                findings.append(f"{name}: type `{m.group(1)}` is not declared in fyrox_hip_sys.rs")
    for f in sorted(set(findings)):
        print(f)
    print(f"{len(set(findings))} finding(s) in {len(SHIM)} file(s)")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
