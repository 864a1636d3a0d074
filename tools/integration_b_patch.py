#!/usr/bin/env python3
"""Materialise INTEGRATION.md section B in a scratch directory (TEST INFRASTRUCTURE; nothing lands in the repo).

  integration_b_patch.py <reference AD-Census dir> <INTEGRATION.md> <out dir>

writes  <out>/ADCensusStereo.h           the reference's header (ADCensusStereo.h:14-95) with the two additions section B names:
                                         `struct adc_handle;` in front of the class, `adc_handle* gpu_ = nullptr;` as a member
        <out>/integration_b_patch.inc    the ```cpp block of section B, verbatim
tests/stubs/integration_b.cpp includes both: the documented patch is compiled exactly as it is printed in the document.
"""
import re
import sys


def main():
    ref, doc, out = sys.argv[1:4]
    hdr = open(ref + "/ADCensusStereo.h", "rb").read().decode("latin-1")  # (GBK comments: bytes pass through untouched)
    assert "class ADCensusStereo" in hdr and "bool is_initialized_;" in hdr
    hdr = hdr.replace("class ADCensusStereo", "struct adc_handle;\nclass ADCensusStereo", 1)
    hdr = hdr.replace("bool is_initialized_;", "bool is_initialized_;\n\tadc_handle* gpu_ = nullptr;", 1)
    open(out + "/ADCensusStereo.h", "wb").write(hdr.encode("latin-1"))
    text = open(doc, encoding="utf-8").read()
    sec = text[text.index("## B. "):]
    block = re.search(r"```cpp\n(.*?)```", sec, re.S).group(1)
    open(out + "/integration_b_patch.inc", "w", encoding="utf-8").write(block)


if __name__ == "__main__":
    main()
