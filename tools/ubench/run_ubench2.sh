#!/bin/bash
# every pattern in its own process under a short timeout
cd "$(dirname "$0")"
for t in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 100 101 102; do
  timeout 25 ./ubench2 $t || echo "test $t: rc=$? (timeout/failed)"
done
