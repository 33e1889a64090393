python -m pytest tests -q -m gpu --tb=short -q 2>&1 | tail -60
cp gpurun_out/measured_errors.json gpurun_out/measured_errors_full.json
cp gpurun_out/parity_exceptions.md gpurun_out/parity_exceptions_full.md
