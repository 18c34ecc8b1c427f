"""Column names shared by the device metric kernels and the host-side tables (importable without the HIP library)."""
OBJ_METRIC_NAMES = ('MCE', 'OCE', 'MCE2', 'ADD', 'ADDS', 'ADD01d', 'ADDS01d', 'REP', 'REP5', 'CD',
                    'FSCORE@2mm', 'FSCORE@5mm', 'FSCORE@10mm', 'FSCORE@2cm', 'FSCORE@5cm', 'FSCORE@10cm')
