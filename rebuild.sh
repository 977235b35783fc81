#!/bin/bash
# rebuild the native libraries in-tree (used during development; __graft_entry__.build() does the same)
cd "$(dirname "$0")" && python -c "
import importlib
importlib.import_module('gadget-leicester_amd').build()
" 2>&1 | grep -E "rror|warning: unused" -A6 | head -40
ls -la "$(dirname "$0")"/gadget-leicester_amd/*.so
