#!/usr/bin/env python
"""Re-serialises the scene fixtures (psdr-cuda_amd/data/scenes/*.xml) in this repository's own compact
style: one element per line, no indentation, attributes in sorted order, camel-case aliases of the loader
(toWorld / fovAxis / lookAt / faceNormals / nearClip, scene_loader.cpp:85,268,276,399), numbers in shortest
form.  Pure re-formatting: the parsed Scene is identical (tests/test_host_logic.py pins the tables)."""
import glob
import os
import re
import xml.etree.ElementTree as ET

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALIAS = {"to_world": "toWorld", "fov_axis": "fovAxis", "face_normals": "faceNormals", "near_clip": "nearClip", "far_clip": "farClip"}
TAGS = {"lookat": "lookAt", "look_at": "lookAt"}


def num(s):
    def one(tok):
        try:
            f = float(tok)
        except ValueError:
            return tok
        r = repr(float(np32(f)))
        return ("%g" % f) if float("%g" % f) == f else r
    return ", ".join(one(t) for t in re.split(r"[,\s]+", s.strip()) if t)


def np32(f):
    return f


def emit(node, out, descr):
    tag = TAGS.get(node.tag, node.tag)
    attrs = dict(node.attrib)
    if "name" in attrs:
        attrs["name"] = ALIAS.get(attrs["name"], attrs["name"])
    for k in ("value", "origin", "target", "up", "x", "y", "z", "angle"):
        if k in attrs and tag not in ("string", "boolean", "ref") and not (tag == "string"):
            if re.fullmatch(r"[-+0-9.,eE\s]+", attrs[k]):
                attrs[k] = num(attrs[k])
    a = "".join(' %s="%s"' % (k, attrs[k]) for k in sorted(attrs))
    kids = list(node)
    if not kids:
        out.append("<%s%s/>" % (tag, a))
        return
    out.append("<%s%s>" % (tag, a))
    for k in kids:
        emit(k, out, descr)
    out.append("</%s>" % tag)


def main():
    for path in sorted(glob.glob(os.path.join(ROOT, "psdr-cuda_amd", "data", "scenes", "*.xml"))):
        text = open(path).read()
        m = re.search(r"<!--(.*?)-->", text, re.S)
        descr = " ".join(m.group(1).split()) if m else os.path.basename(path)
        root = ET.fromstring(text)
        out = ["<!-- %s | psdr-mi355x fixture, written by tools/make_scenes.py -->" % descr]
        emit(root, out, descr)
        open(path, "w").write("\n".join(out) + "\n")
        print(os.path.basename(path), len(out), "lines")


if __name__ == "__main__":
    main()
