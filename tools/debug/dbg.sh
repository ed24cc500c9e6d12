python tools/debug/concurrent_forward.py 16 fp16 3x128,2x192 30
python tools/debug/concurrent_forward.py 16 fp32 3x128,2x192 30
python tools/debug/concurrent_forward.py 16 fp16 3x128,3x128 30
python tools/debug/concurrent_forward.py 128 bf16 2x128,3x192 15
python tools/debug/concurrent_forward.py 128 bf16 1x512,1x512 15
python tools/debug/concurrent_forward.py 16 fp16 3x128,2x192 30 STORM_CONV_VARIANT=0
