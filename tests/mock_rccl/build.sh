#!/bin/bash
# builds the test-only mock of librccl (see mock_rccl.cpp) next to its source
cd "$(dirname "$0")" && hipcc -O2 -fPIC -shared -std=c++17 mock_rccl.cpp -o librccl_mock.so -lrt && echo built tests/mock_rccl/librccl_mock.so
