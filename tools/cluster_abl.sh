#!/bin/bash
# timing ablations of the cluster kernel (make ABLATE=1 library): where does a step's time go
for abl in ${ABLS:-0 1 2 9 25}; do
  FNSSL_CLUSTER_ABL=$abl timeout 100 python tools/cluster_check.py time 2>&1 | grep "cluster\|equal" | sed "s/^/ABL=$abl  /"
done
