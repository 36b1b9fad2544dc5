#!/usr/bin/env python3
"""Concatenate the device template's parts (kernels/ORDER) and wrap the text into a C++ raw string
literal (n2_kernels_embed.h) for rmt_n2.cpp.  `embed.py --cat` prints the template itself."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def parts():
    out = []
    for line in open(os.path.join(HERE, "kernels", "ORDER")):
        name = line.split("#", 1)[0].strip()
        if name:
            out.append(os.path.join(HERE, "kernels", name))
    return out


def template():
    return "".join(open(p).read() for p in parts())


if __name__ == "__main__":
    text = template()
    if sys.argv[1] == "--cat":
        sys.stdout.write(text)
    else:
        assert ')RMTSRC"' not in text
        with open(sys.argv[1], "w") as f:
            f.write('R"RMTSRC(' + text + ')RMTSRC"\n')
