bash tools/shapes.sh c3 bench kt pmc 2>&1 | grep -v "^$" | tail -12
bash tools/shapes.sh c5 bench kt pmc 2>&1 | grep -v "^$" | tail -12
