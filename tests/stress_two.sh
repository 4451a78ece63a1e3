#!/bin/bash
# usage: two.sh "<env assignments>" : two concurrent determinism runs, prints their summaries
(env $1 timeout -k 10 300 python tests/stress_determinism.py 30 > /tmp/a.txt 2>&1 &)
env $1 timeout -k 10 300 python tests/stress_determinism.py 30 2>&1 | tail -1
sleep 4; tail -1 /tmp/a.txt
