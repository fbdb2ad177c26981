#!/bin/bash
# Build the HIP extension + oracle from anywhere.
cd "$(dirname "$0")/.." && python -c "import __graft_entry__ as g; g.build()"
