python -m pytest tests -m gpu -q -x --durations=6 2>&1 | tail -14
