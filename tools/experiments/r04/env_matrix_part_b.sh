#!/bin/bash
ENVM_FROM=1 ENVM_TO=2 bash tools/env_matrix.sh 2>&1 | grep -v "^# HEAD"
ENVM_FROM=5 ENVM_TO=5 bash tools/env_matrix.sh 2>&1 | grep -v "^# HEAD"
ENVM_FROM=12 bash tools/env_matrix.sh 2>&1
