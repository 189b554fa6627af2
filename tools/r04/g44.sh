#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\(HIT\|MISS\|RDREQ\|EA0_RDREQ\)[A-Za-z0-9_]*" | sort -u | head -30
