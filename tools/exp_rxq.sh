# GPU box: the light-mix and survey-mix banks with and without torch owning the device first
for mix in light survey; do
for mode in plain torch; do
  if [ $mode = torch ]; then export KG_TOOL_TORCH=1; else unset KG_TOOL_TORCH; fi
  echo "== $mix $mode"; python3 tools/time_rxbank.py $mix 128 60 2>&1 | tail -2
done; done
