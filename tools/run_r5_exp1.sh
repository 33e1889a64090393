GRAFX_PARITY_UPDATE=replace python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -25
cp tests/parity_exceptions_allowed.json gpurun_out/allowed_replace.json
cp gpurun_out/parity_exceptions.md gpurun_out/parity_exceptions_full.md
cp gpurun_out/measured_errors.json gpurun_out/measured_errors_full.json
