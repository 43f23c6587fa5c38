#!/bin/bash
# the GPU suite three times in a row on one box: anything flaky?
for i in 1 2 3; do timeout 1500 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -2; done
